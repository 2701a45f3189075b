import sys, pytest
sys.path.insert(0, '.')
from imageanalysis_pipeline_amd import _lib
keep = [k for k in _lib.SIGNATURES if not k.startswith(("iap_reg","iap_mip","iap_apply","iap_im_reg"))]
_lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if k in keep}
sys.exit(pytest.main(["tests/test_bg_gpu.py", "-m", "gpu", "-x", "-q", "--timeout", "120", "--timeout-method", "thread"]))
