cd $GRAFT_REPO_ROOT
echo "== 1 pytest new"; timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "any_device_address" 2>&1 | tail -3
echo "== 2 pytest old lib"; TIKTOKEN_AMD_LIB=$PWD/tiktoken_amd/csrc/variants/libtiktoken_amd_ef0.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "any_device_address" 2>&1 | tail -3
echo "== 3 direct call"; cd tests; timeout 300 python -c "
import sys; sys.path.insert(0,'..')
import test_gpu_parity as t; t.test_text_at_any_device_address(); print('direct ok')" 2>&1 | tail -3
echo "== 4 pytest, torch first"; cd ..; timeout 300 python -c "
import torch; torch.zeros(1).cuda()
import pytest, sys; sys.exit(pytest.main(['tests/test_gpu_parity.py','-m','gpu','-q','-x','-k','any_device_address']))" 2>&1 | tail -3
