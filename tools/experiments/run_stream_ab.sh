for l in st stov; do
python - <<PY
import sys, os
sys.path.insert(0, "sfmnext-impl_amd")
from sqd import lib as _l
_l.SO_PATH = os.path.abspath("tools/bin/libsqd_$l.so"); _l.needs_build = lambda: False
import pytest
sys.exit(pytest.main(["tests/test_gpu_photometric.py", "tests/test_gpu_golden_replay.py", "tests/test_gpu_stereo.py", "-x", "-q", "-p", "no:cacheprovider"]))
PY
done 2>&1 | grep -E "passed|failed" 
for i in 1 2; do echo -n "tile: "; python tools/bench_fused.py --which fwd --iters 300 2>&1 | tail -1; for l in st stov; do echo -n "$l: "; python tools/bench_fused.py --which fwd --iters 300 --lib tools/bin/libsqd_$l.so 2>&1 | tail -1; done; done
