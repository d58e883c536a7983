python -m pytest tests/test_gpu_kernels.py -q -k "roi_pool" 2>&1 | tail -3
for sh in "63 63 4000" "99 165 8000" "76 114 4000" "150 200 4000" "125 167 4000"; do
    python tools/roi_fwd_forms.py one $sh 2>&1 | grep -v amdgpu 
    SW_ROI_SPARSE_CB4=1 python tools/roi_fwd_forms.py one $sh 2>&1 | grep -v amdgpu | sed "s/^/cb4 /"
    SW_ROI_SPARSE_CB4=1 SW_ROI_FWD_WGS=768 python tools/roi_fwd_forms.py one $sh 2>&1 | grep -v amdgpu | sed "s/^/cb4 wgs768 /"
done
