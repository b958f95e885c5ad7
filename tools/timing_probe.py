import sys, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import synthetic_image
from dlimgedit_amd import api, weights as W
from dlimgedit_amd.sam_config import get_config
cfg = get_config("vit_b")
with tempfile.TemporaryDirectory() as d:
    W.write_synthetic_model_dir(d, cfg, seed=0)
    env = api.Environment(api.Options(api.Backend.gpu, d))
    views = [api.ImageView(synthetic_image(i), api.Channels.rgba) for i in range(8)]
    for _ in range(4):
        segs = api.Segmentation.process_batch(views, env)
    for _ in range(3):
        s = api.Segmentation.process(views[0], env)
