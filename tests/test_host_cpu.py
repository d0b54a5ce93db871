"""CPU-only checks: the C-ABI library loads and exports every symbol include/vitk.h declares,
the host-side mirror of the reference interface (class names, signatures, state_dict keys,
presets, checkpoint loader, LR schedules) and the failure mode without a GPU."""
import ctypes
import inspect
import math
import os
import re

import numpy as np
import pytest
import torch

from oracle import synth, vit_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vitk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vitk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import vitk
    lib = ctypes.CDLL(vitk.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vitk.h but not exported"
    assert vitk.lib.vitk_version() == 100
    assert b"constraint" in vitk.lib.vitk_error_string(-1)


def test_binding_prototypes_match_header_arity():
    import vitk
    text = open(os.path.join(ROOT, "include", "vitk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, argtypes in vitk.PROTOTYPES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        nargs = len([a for a in m.group(1).split(",") if a.strip()])
        assert nargs == len(argtypes), (name, nargs, len(argtypes))


def test_epilogue_struct_layout_matches_c():
    import vitk
    assert ctypes.sizeof(vitk.EpilogueArgs) == 64
    assert vitk.EpilogueArgs.seed.offset == 56 and vitk.EpilogueArgs.npatch.offset == 48


def test_module_interface_mirrors_reference():
    """Constructor / forward signatures and state_dict keys (reference modeling.py:50-279)."""
    from models import config as cfgmod
    from models.ViT import modeling as M
    cfg = cfgmod.get_config("ViT-B_16")
    assert list(inspect.signature(M.Attention.__init__).parameters) == ["self", "config", "vis"]
    assert list(inspect.signature(M.Mlp.__init__).parameters) == ["self", "config"]
    assert list(inspect.signature(M.Embeddings.__init__).parameters) == ["self", "config", "img_size", "in_channels"]
    assert list(inspect.signature(M.Block.__init__).parameters) == ["self", "config", "vis"]
    assert list(inspect.signature(M.Encoder.__init__).parameters) == ["self", "config", "vis"]
    assert list(inspect.signature(M.Attention.forward).parameters)[:2] == ["self", "hidden_states"]
    assert list(inspect.signature(M.VisionTransformer.forward).parameters) == ["self", "x", "labels", "task"]
    model = M.VisionTransformer(cfg, 224, num_classes=10, zero_head=True)
    want = [n for n, _ in synth.param_shapes("B_16", 224, 10)]
    got = dict(model.state_dict())
    assert sorted(got) == sorted(want)
    for n, shape in synth.param_shapes("B_16", 224, 10):
        assert tuple(got[n].shape) == shape and got[n].dtype == torch.float32, n
    assert sum(p.numel() for p in model.parameters()) == 85806346
    blk = model.transformer.encoder.layer[0]
    for attr in ("hidden_size", "attn", "ffn", "attention_norm", "ffn_norm"):
        assert hasattr(blk, attr)
    assert model.transformer.embeddings.hybrid is False


def test_init_matches_reference_distribution():
    """Mlp init: xavier-uniform weights, N(0, 1e-6) biases (reference modeling.py:107-111);
    position embeddings and class token start at zero (:148-149)."""
    from models import config as cfgmod
    from models.ViT import modeling as M
    torch.manual_seed(0)
    m = M.VisionTransformer(cfgmod.get_config("ViT-B_16"), 224, num_classes=10)
    fc1 = m.transformer.encoder.layer[3].ffn.fc1
    bound = math.sqrt(6.0 / (768 + 3072))
    assert float(fc1.weight.abs().max()) <= bound and float(fc1.weight.abs().max()) > 0.98 * bound
    assert float(fc1.bias.abs().max()) < 1e-5
    assert float(m.transformer.embeddings.position_embeddings.abs().max()) == 0.0
    assert float(m.transformer.embeddings.cls_token.abs().max()) == 0.0


def test_presets():
    from models import config as cfgmod
    for name in ("ViT-B_16", "ViT-B_16_MultiHead", "ViT-B_16_RKR", "ViT-B_16_RKRwoRG", "ViT-B_16_RKRwoSFG",
                 "ViT-B_16_RKRTSN", "ViT-B_16_PB", "ViT-B_16_RKRPB", "ViT-B_16_RKRPBG", "ViT-B_16_PBG"):
        c = cfgmod.get_config(name)
        assert c.hidden_size == 768 and c.transformer["mlp_dim"] == 3072
        assert c.transformer["num_heads"] == 12 and c.transformer.num_layers == 12
        assert c.transformer["attention_dropout_rate"] == 0.0 and c.transformer["dropout_rate"] == 0.1
        assert c.patches["size"] == (16, 16) and c.patches.get("grid") is None and c.classifier == "token"
    assert cfgmod.get_config("ViT-B_16_RKR").RG is True and cfgmod.get_config("ViT-B_16_RKR").K == 2
    assert cfgmod.get_config("ViT-B_16_PBG").lamb == 0.25
    l16 = cfgmod.get_config("ViT-L_16")
    assert (l16.hidden_size, l16.transformer.mlp_dim, l16.transformer.num_heads, l16.transformer.num_layers) == \
        (1024, 4096, 16, 24)
    r50 = cfgmod.get_config("R50-ViT-B_16")
    assert r50.patches.get("grid") == (14, 14) and r50.resnet.num_layers == (3, 4, 9)
    sw = cfgmod.get_config("Swin", "cifar100")
    assert sw.embed_dim == 96 and sw.depths == [2, 2] and sw.window_size == 7
    assert cfgmod.get_config("ResNet18_RKR").rkr_scale == 1e-1
    with pytest.raises(KeyError):
        cfgmod.get_config("nope")


def _synth_npz(geom, img, ncls, seed):
    """Same generator as oracle/gen_golden.py:synth_npz (JAX checkpoint key layout)."""
    g = synth.GEOMS[geom]
    D, F_, P, H = g["hidden"], g["mlp"], g["patch"], g["heads"]
    rng = np.random.default_rng(seed)
    n = (img // P) ** 2
    r = lambda *s: rng.standard_normal(s).astype(np.float32) * 0.05  # noqa: E731
    w = {"embedding/kernel": r(P, P, 3, D), "embedding/bias": r(D), "cls": r(1, 1, D),
         "Transformer/encoder_norm/scale": 1 + r(D), "Transformer/encoder_norm/bias": r(D),
         "Transformer/posembed_input/pos_embedding": r(1, n + 1, D),
         "head/kernel": r(D, ncls), "head/bias": r(ncls)}
    for i in range(g["layers"]):
        root = f"Transformer/encoderblock_{i}/"
        for nm in ("query", "key", "value"):
            w[root + f"MultiHeadDotProductAttention_1/{nm}/kernel"] = r(D, H, D // H)
            w[root + f"MultiHeadDotProductAttention_1/{nm}/bias"] = r(H, D // H)
        w[root + "MultiHeadDotProductAttention_1/out/kernel"] = r(H, D // H, D)
        w[root + "MultiHeadDotProductAttention_1/out/bias"] = r(D)
        w[root + "MlpBlock_3/Dense_0/kernel"] = r(D, F_)
        w[root + "MlpBlock_3/Dense_0/bias"] = r(F_)
        w[root + "MlpBlock_3/Dense_1/kernel"] = r(F_, D)
        w[root + "MlpBlock_3/Dense_1/bias"] = r(D)
        w[root + "LayerNorm_0/scale"] = 1 + r(D)
        w[root + "LayerNorm_0/bias"] = r(D)
        w[root + "LayerNorm_2/scale"] = 1 + r(D)
        w[root + "LayerNorm_2/bias"] = r(D)
    return w


@pytest.mark.parametrize("tag,img,zero_head", [("same", 32, False), ("resize", 48, True)])
def test_load_from_matches_reference(golden_dir, tag, img, zero_head):
    """JAX-checkpoint loader incl. the position-embedding grid resize, against what the reference's
    load_from produced from the same synthetic checkpoint (bit-exact: pure data movement + zoom)."""
    from models import config as cfgmod
    from models.ViT.modeling import VisionTransformer
    g = np.load(os.path.join(golden_dir, "loadfrom_tiny.npz"))
    model = VisionTransformer(cfgmod.get_tiny_config(), img, num_classes=10, zero_head=zero_head)
    model.load_from(_synth_npz("tiny", 32, 10, seed=21))
    for k, v in model.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), g[f"{tag}.{k}"], err_msg=k)


def test_schedules_match_oracle():
    from utils import scheduler as S
    p = [torch.nn.Parameter(torch.zeros(1))]
    for cls, ref, kw in ((S.WarmupCosineSchedule, O.warmup_cosine, dict(warmup_steps=3, t_total=11)),
                         (S.WarmupLinearSchedule, O.warmup_linear, dict(warmup_steps=3, t_total=11))):
        opt = torch.optim.SGD(p, lr=2.0)
        sch = cls(opt, **kw)
        for step in range(14):
            assert abs(opt.param_groups[0]["lr"] - 2.0 * ref(step, 3, 11)) < 1e-12, (cls.__name__, step)
            opt.step()
            sch.step()
    opt = torch.optim.SGD(p, lr=2.0)
    sch = S.WarmupConstantSchedule(opt, warmup_steps=4)
    lrs = []
    for _ in range(6):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    assert lrs == [0.0, 0.5, 1.0, 1.5, 2.0, 2.0]
    assert S.ConstantLRSchedule(torch.optim.SGD(p, lr=1.0)).get_last_lr() == [1.0]


def test_forward_without_gpu_fails_loudly():
    import vitk
    from models import config as cfgmod
    from models.ViT.modeling import VisionTransformer
    model = VisionTransformer(cfgmod.get_tiny_config(), 32, num_classes=10)
    with pytest.raises(vitk.VitkError):
        model(torch.zeros(1, 3, 32, 32))


def test_r50_hybrid_preset_builds_the_stem():
    """R50-ViT-B_16 (reference _configs.py:224-232): 3/4/9 bottleneck units, 1024-channel feature map
    on a 14x14 grid, 1x1 patch projection."""
    from models import config as cfgmod
    from models.ViT.modeling import Embeddings
    from models.ViT.modeling_resnet import ResNetV2
    emb = Embeddings(cfgmod.get_config("R50-ViT-B_16"), 224)
    assert emb.hybrid and isinstance(emb.hybrid_model, ResNetV2)
    assert [len(b) for b in emb.hybrid_model.body] == [3, 4, 9]
    assert emb.patch_embeddings.in_channels == 1024 and emb.patch_embeddings.kernel_size == (1, 1)
    assert emb.position_embeddings.shape == (1, 197, 768)


def test_bench_started_plainly_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts the ranks itself as child processes (reference
    train.py:652-666 relies on an external launcher); rehearsed on the CPU with --dry-run (gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["allreduce_sum"] == out["expected"] == 3.0
