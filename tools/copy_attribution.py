"""Where the __amd_rocclr_copyBuffer blit kernels of a profiled run come from (VERDICT r05, weak point 12): run under
`rocprofv3 --kernel-trace --stats` with N = 0 and N = 50 iterations of process + compute_mask behind the model load; the
difference of the copy-kernel counts / 50 is what one request costs, the N = 0 count is the model load.
    rocprofv3 --kernel-trace --stats -d gpurun_out/copies0 -o c0 -- python3 tools/copy_attribution.py 0
    rocprofv3 --kernel-trace --stats -d gpurun_out/copies50 -o c50 -- python3 tools/copy_attribution.py 50"""
import sys
import tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import synthetic_image                     # noqa: E402
from dlimgedit_amd import api, weights as W              # noqa: E402
from dlimgedit_amd.sam_config import get_config          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = get_config("vit_b")
with tempfile.TemporaryDirectory() as d:
    W.write_synthetic_model_dir(d, cfg, seed=0)
    env = api.Environment(api.Options(api.Backend.gpu, d))
    view = api.ImageView(synthetic_image(0), api.Channels.rgba)
    api.Segmentation.process(view, env).compute_mask(api.Point(512, 512))        # model load + first launches
    for _ in range(n):
        api.Segmentation.process(view, env).compute_mask(api.Point(512, 512))
    env.close()
