"""Converts a SAM checkpoint to the DLW file the MI355X build loads.

    python tools/convert_checkpoint.py sam_vit_b_01ec64.pth vit_b <model_directory>        # Meta .pth
    python tools/convert_checkpoint.py model.safetensors     vit_h <model_directory> --hf   # Hugging Face SamModel

Writes <model_directory>/segmentation/sam_<variant>.dlw.  Needs torch (and safetensors for --hf files)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import weights as W            # noqa: E402
from dlimgedit_amd.sam_config import get_config   # noqa: E402


def main():
    if len(sys.argv) < 4:
        raise SystemExit(__doc__)
    src, variant, out_dir = sys.argv[1], sys.argv[2], sys.argv[3]
    cfg = get_config(variant)
    if src.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(src)
    else:
        import torch
        sd = torch.load(src, map_location="cpu", weights_only=True)
    params = W.from_hf_state_dict(cfg, sd) if "--hf" in sys.argv else W.from_meta_state_dict(cfg, sd)
    path = W.save_weights(Path(out_dir) / "segmentation" / W.weight_file_name(cfg), cfg, params)
    print(f"wrote {path} ({path.stat().st_size / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
