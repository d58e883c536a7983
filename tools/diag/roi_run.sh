python -m pytest tests/test_gpu_kernels.py -q -k "roi_pool" 2>&1 | tail -3
python tools/roi_fwd_forms.py 2>&1 | grep -v amdgpu
