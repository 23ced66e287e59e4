#!/bin/bash
# Round 5: the latency tables on the final code (small calls, mid-size calls, web text per call size, long runs, decode path)
TAG=r05
mkdir -p gpurun_out
timeout 60 python tools/mid_corpus.py > gpurun_out/${TAG}_mid_calls_corpus.txt 2>&1; tail -12 gpurun_out/${TAG}_mid_calls_corpus.txt
timeout 40 python tools/stress_repeats.py o200k_shaped > gpurun_out/${TAG}_long_runs.txt 2>&1; tail -5 gpurun_out/${TAG}_long_runs.txt
timeout 40 python tools/mid_call.py > gpurun_out/${TAG}_mid_calls.txt 2>&1; tail -6 gpurun_out/${TAG}_mid_calls.txt
timeout 50 python tools/small_call.py > gpurun_out/${TAG}_small_calls.txt 2>&1; tail -8 gpurun_out/${TAG}_small_calls.txt
timeout 30 python tools/decode_path.py > gpurun_out/${TAG}_decode_path.txt 2>&1; tail -4 gpurun_out/${TAG}_decode_path.txt
