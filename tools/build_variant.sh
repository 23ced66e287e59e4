#!/bin/bash
# usage: tools/build_variant.sh NAME [-DTKF_HOT_BITS=11 -DTKF_OCC=3 ...]
# Builds the product library with other compile-time parameters into tiktoken_amd/csrc/variants/libtiktoken_amd_NAME.so (experiments only:
# $TIKTOKEN_AMD_LIB selects it at run time; the shipped library is the Makefile's).
set -e
name=$1; shift
cd "$(dirname "$0")/../tiktoken_amd/csrc"
mkdir -p variants/obj_$name
o=variants/obj_$name
g++ -O2 -std=c++17 -fPIC -Wall "$@" -c tk_tables.cpp -o $o/tk_tables.o &
g++ -O2 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c tk_pattern.cpp -o $o/tk_pattern.o &
g++ -O2 -std=c++17 -fPIC -Wall "$@" -c tk_regex.cpp -o $o/tk_regex.o &
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed "$@" -c tk_api.hip -o $o/tk_api.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $o/tk_api.o $o/tk_tables.o $o/tk_pattern.o $o/tk_regex.o -o variants/libtiktoken_amd_$name.so
rm -rf $o
echo built variants/libtiktoken_amd_$name.so
