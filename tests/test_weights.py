"""Weight inventory, seeded generator and the DLW file format (dlimgedit_amd/weights.py)."""
import numpy as np
import pytest
from pathlib import Path

from dlimgedit_amd import weights as W
from dlimgedit_amd.sam_config import CONFIGS, get_config


def test_counter_generator_is_a_pure_function():
    a = W.counter_uniform(7, "enc.patch.w", 1000)
    b = W.counter_uniform(7, "enc.patch.w", 2000)[:1000]
    assert np.array_equal(a, b)
    assert not np.array_equal(a, W.counter_uniform(8, "enc.patch.w", 1000))
    assert not np.array_equal(a, W.counter_uniform(7, "enc.patch.b", 1000))
    assert a.dtype == np.float32 and a.min() >= -1 and a.max() < 1
    # pinned values: the stream must never change (golden fixtures depend on it)
    assert [float(v).hex() for v in W.counter_uniform(0, "x", 3)] == ["0x1.e684200000000p-3", "0x1.292d0c0000000p-1", "0x1.82a6f80000000p-2"]
    assert abs(float(a.mean())) < 0.1 and 0.5 < float(a.std()) < 0.65


def test_param_inventory_matches_published_sizes():
    """ViT-B / ViT-H encoder parameter counts of the public SAM release (89.67 M / 637.0 M)."""
    def enc_params(cfg):
        return sum(int(np.prod(s)) for n, s, _ in W.param_specs(cfg) if n.startswith("enc."))
    assert abs(enc_params(get_config("vit_b")) / 1e6 - 89.67) < 0.01
    assert abs(enc_params(get_config("vit_h")) / 1e6 - 637.0) < 0.1


def test_flop_accounting_matches_survey():
    assert abs(get_config("vit_b").encoder_flops() / 1e9 - 941.7) < 0.5
    assert abs(get_config("vit_h").encoder_flops() / 1e9 - 5666) < 5


def test_dlw_round_trip(tmp_path):
    cfg = get_config("vit_test")
    params = W.synthetic_weights(cfg, 3)
    path = W.save_weights(tmp_path / "segmentation" / W.weight_file_name(cfg), cfg, params)
    meta, back = W.load_weights(path)
    assert meta == {"embed_dim": 128, "depth": 2, "num_heads": 2, "mlp_dim": 512, "global_attn_indexes": (1,)}
    assert set(back) == set(params)
    for k in params:
        assert np.array_equal(back[k], params[k]), k
    assert path.read_bytes()[:8] == b"DLIMGSAM"


def test_save_rejects_incomplete_or_misshapen(tmp_path):
    cfg = get_config("vit_test")
    params = W.synthetic_weights(cfg, 3)
    bad = dict(params)
    del bad["dec.iou_token"]
    with pytest.raises(ValueError, match="missing tensors"):
        W.save_weights(tmp_path / "a.dlw", cfg, bad)
    bad = dict(params)
    bad["enc.pos"] = bad["enc.pos"][:10]
    with pytest.raises(ValueError, match="shape"):
        W.save_weights(tmp_path / "b.dlw", cfg, bad)


def test_hf_mapping_is_a_bijection_on_our_inventory():
    for name in ("vit_test", "vit_b"):
        cfg = CONFIGS[name]
        specs = {n: s for n, s, _ in W.param_specs(cfg)}
        sd = W.to_hf_state_dict(cfg, {n: np.zeros(s, np.float32) for n, s in specs.items()})
        back = W.from_hf_state_dict(cfg, sd)
        assert set(back) == set(specs)
        assert all(tuple(back[n].shape) == tuple(specs[n]) for n in specs)


def test_meta_checkpoint_mapping_round_trips():
    """Meta `segment_anything` key names <-> ours (tools/convert_checkpoint.py path)."""
    cfg = CONFIGS["vit_test"]
    params = W.synthetic_weights(cfg, 5)
    sd = W.to_meta_state_dict(cfg, params)
    assert "image_encoder.blocks.1.attn.rel_pos_h" in sd and "mask_decoder.output_upscaling.3.weight" in sd
    sd["prompt_encoder.mask_downscaling.0.weight"] = np.zeros((4, 1, 2, 2), np.float32)    # unused branch is ignored
    back = W.from_meta_state_dict(cfg, sd)
    assert set(back) == set(params)
    for k in params:
        assert np.array_equal(back[k], params[k]), k
    with pytest.raises(ValueError, match="wrong variant"):
        W.from_meta_state_dict(CONFIGS["vit_b"], sd)


@pytest.mark.parametrize("fmt", ["pth", "safetensors"])
def test_convert_checkpoint_tool(tmp_path, fmt):
    """tools/convert_checkpoint.py end to end: a Meta-style .pth / an HF-style .safetensors file in, the DLW file the
    library loads out."""
    import subprocess
    import sys
    import torch
    cfg = CONFIGS["vit_test"]
    params = W.synthetic_weights(cfg, 3)
    if fmt == "pth":
        sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in W.to_meta_state_dict(cfg, params).items()}
        src = tmp_path / "sam_test.pth"
        torch.save(sd, src)
        extra = []
    else:
        from safetensors.torch import save_file
        # HF ties two names to one tensor; a file holds them as separate copies
        sd = {k: torch.from_numpy(np.array(v, copy=True)) for k, v in W.to_hf_state_dict(cfg, params).items()}
        src = tmp_path / "model.safetensors"
        save_file(sd, str(src))
        extra = ["--hf"]
    tool = Path(__file__).resolve().parent.parent / "tools" / "convert_checkpoint.py"
    r = subprocess.run([sys.executable, str(tool), str(src), "vit_test", str(tmp_path / "models"), *extra],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cfg2, back = W.load_weights(tmp_path / "models" / "segmentation" / W.weight_file_name(cfg))
    assert cfg2["embed_dim"] == cfg.embed_dim and cfg2["depth"] == cfg.depth
    for name, value in params.items():
        assert np.array_equal(back[name], value), name


def test_values_outside_the_f16_range_are_refused_when_a_model_is_written(tmp_path):
    """No real checkpoint has run here, and f16 operands overflow silently on the device (65504): the writer -- which
    tools/convert_checkpoint.py goes through -- refuses what cannot be represented, by tensor name."""
    cfg = get_config("vit_test")
    params = W.synthetic_weights(cfg, 3)
    bad = dict(params)
    bad["enc.L0.fc2.w"] = params["enc.L0.fc2.w"].copy()
    bad["enc.L0.fc2.w"][3, 5] = 7.0e4
    with pytest.raises(ValueError, match=r"enc\.L0\.fc2\.w.*f16 range"):
        W.save_weights(tmp_path / "a.dlw", cfg, bad)
    W.save_weights(tmp_path / "a2.dlw", cfg, bad, allow_out_of_range=True)      # the tests' way to reach the C++ loader
    bad = dict(params)
    bad["enc.L1.ln1.b"] = params["enc.L1.ln1.b"].copy()
    bad["enc.L1.ln1.b"][0] = np.nan
    with pytest.raises(ValueError, match="non-finite"):
        W.save_weights(tmp_path / "b.dlw", cfg, bad, allow_out_of_range=True)
    # fp32-only tensors may be large: the residual stream and the token side of the decoder are not f16 operands
    ok = dict(params)
    ok["enc.L0.fc2.b"] = params["enc.L0.fc2.b"] + 1.0e5
    W.save_weights(tmp_path / "c.dlw", cfg, ok)
    assert W.f16_operand("enc.L0.qkv.w") and W.f16_operand("enc.L1.rel_h") and not W.f16_operand("enc.L0.ln1.w")
