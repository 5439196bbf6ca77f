"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI by the
btsbot_amd modules, against the CPU oracle and the committed reference-wrapper goldens.

Tolerances (the reference's own notion of "same output" is rtol 1e-4 / atol 1e-5 on fp32 logits,
/root/reference/btsbot/to_onnx.py:135-137; the north star asks for scores within 1e-4).  The two modes that claim the
north star are held to ITS constant, |dscore| <= 1e-4 (f16x2) and 1e-5 (f32); the plain 16-bit modes do not meet it
with these stress weights and are held to at most twice what is measured on these seeds, so a numerical regression
fails the suite:
  f32   mode: |dlogit| <= 1e-4 * max(1, max|logit|)  and  |dscore| <= 1e-5   (measured 7e-6 / 2e-6)
  f16x2 mode: |dscore| <= 1e-4, the north-star tolerance itself (split operands: f16 head + f16 remainder; measured
             6.9e-5 at B = 1024 with layer-scale gamma ~ 1, of which the operands left plain f16 -- the depthwise
             input map and the pointwise filters of stages 0-1, the stem -- own all of it: tools/error_budget2.py)
  f16  mode: |dscore| <= 3e-4   (measured 1.4e-4 at B = 39, 1.7e-4 at B = 1024 with layer-scale gamma ~ 1;
             5e-5 with gamma ~ 0.1, test_f16_meets_1e4_at_trained_like_layer_scale)
  bf16 mode: |dscore| <= 2.5e-3 (measured 1.0e-3 .. 1.3e-3, gamma ~ 1)
What owns the 16-bit error is the operand rounding itself (CPU emulation, tools/error_budget.py: LayerNorm
outputs 9e-5, filters 7e-5, hidden activations 6e-5 of the f16 mode's 1.7e-4), not a kernel: a mode that
feeds 16-bit operands to the matrix pipe cannot do better at gamma ~ 1.
Weights are seeded random with layer-scale gamma ~ 1 (a trained checkpoint has |gamma| << 1, which
damps the low-precision error of every block); no trained checkpoint exists offline.
"""
import os

import numpy as np
import pytest
import torch

from helpers import CONFIGS, seeded_state, build_model, run_model
from btsbot_amd import _lib
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O   # checker only

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
NORTH_STAR = 1e-4                       # BASELINE.json: "scores within 1e-4 of the reference"
TOL_SCORE = {"f32": 1e-5, "f16x2": NORTH_STAR, "f16": 3e-4, "bf16": 2.5e-3, "fp8": 3.5e-2}
# |dlogit| <= TOL_LOGIT_REL * max(1, max|logit|): the score bound alone is vacuous where the logits are large and the
# sigmoid saturated (the MaxViT wirings with seeded weights); at most twice what is measured
# (measured worst over the wirings, tools/logit_tol.py: 1.4e-6 / 2.3e-4 / 9.2e-4 / 5.4e-3 / 5.5e-2)
TOL_LOGIT_REL = {"f32": 1e-4, "f16x2": 4.5e-4, "f16": 1.8e-3, "bf16": 1.1e-2, "fp8": 0.11}


def _oracle(kind, cfg, sd, img, meta):
    with torch.no_grad():
        return O.forward(kind, sd, cfg, img, meta)


def _check(out, ref, prec):
    out = out.cpu()
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert torch.isfinite(out).all()
    ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs().max().item()
    assert ds <= TOL_SCORE[prec], f"{prec}: max|dscore| {ds}"
    scale = max(1.0, ref.abs().max().item())
    dl = (out - ref).abs().max().item()
    assert dl <= TOL_LOGIT_REL[prec] * scale, f"{prec}: max|dlogit| {dl} (scale {scale})"
    return ds


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("prec", ["f32", "f16x2", "bf16", "f16", "fp8"])
def test_forward_matches_oracle(cuda, name, prec):
    kind, cfg = CONFIGS[name]
    sd = seeded_state(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(39, seed=2)             # the example-data batch size
    ref = _oracle(kind, cfg, sd, img, meta)
    m = build_model(kind, cfg, sd, cuda, prec)
    out = run_model(kind, m, img.to(cuda), meta.to(cuda))
    _check(out, ref, prec)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_forward_matches_reference_wrapper_goldens(cuda, name):
    """Committed logits of the reference's own nn.Modules on 8 bundled example alerts."""
    kind, cfg = CONFIGS[name]
    gold = np.load(os.path.join(GOLD, "ref_logits.npz"))
    ex = np.load(os.path.join(GOLD, "example8.npz"))
    sd = seeded_state(kind, cfg, seed=3)
    m = build_model(kind, cfg, sd, cuda, "f32")
    img = torch.from_numpy(ex["triplets"]).to(cuda)
    meta = torch.from_numpy(ex["metadata"]).to(cuda)
    out = run_model(kind, m, img, meta)
    _check(out, torch.from_numpy(gold[f"{name}/example8"]), "f32")


def test_stage_activations_match_oracle(cuda):
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(5, seed=7)
    taps = {}
    with torch.no_grad():
        O.mm_convnext_forward(sd, cfg, img, meta, taps=taps)
    m = build_model(kind, cfg, sd, cuda, "f32")
    m.set_debug_taps(True)
    run_model(kind, m, img.to(cuda), meta.to(cuda))
    for t in ("stem", "stage0", "stage1", "stage2", "stage3"):
        got = m.read_tap(t).cpu()
        ref = taps[t].permute(0, 2, 3, 1).reshape(got.shape)   # NCHW -> NHWC rows
        assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), t


@pytest.mark.parametrize("batch", [1, 7, 8, 9, 64, 2049])
def test_ragged_batches(cuda, batch):
    """Batch sizes around the kernels' grouping factors (2/8 maps per workgroup, 8 alerts per head
    workgroup, 128-row GEMM tiles) and one past the internal 2048-alert chunk."""
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(min(batch, 64), seed=9)
    reps = (batch + img.shape[0] - 1) // img.shape[0]
    img, meta = img.repeat(reps, 1, 1, 1)[:batch], meta.repeat(reps, 1)[:batch]
    ref = _oracle(kind, cfg, sd, img[:64], meta[:64])
    ref = ref.repeat(reps, 1)[:batch]
    m = build_model(kind, cfg, sd, cuda, "f32")
    m._max_chunk = 2048                    # (the default chunk is larger: keep the 2049th alert in a chunk of its own)
    out = run_model(kind, m, img.to(cuda), meta.to(cuda))
    _check(out, ref, "f32")


@pytest.mark.parametrize("batch", [1, 3, 7, 9, 2049])
def test_ragged_batches_split_mode(cuda, batch):
    """The split-operand kernels' partial workgroups (1 / 2 / 4 alerts per workgroup in stages 0 / 1 / 2, 32- and
    64-alert tiles in stage 3, 16 per head workgroup) and one alert past the internal 2048-alert chunk."""
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(min(batch, 64), seed=9)
    reps = (batch + img.shape[0] - 1) // img.shape[0]
    img, meta = img.repeat(reps, 1, 1, 1)[:batch], meta.repeat(reps, 1)[:batch]
    ref = _oracle(kind, cfg, sd, img[:64], meta[:64]).repeat(reps, 1)[:batch]
    m = build_model(kind, cfg, sd, cuda, "f16x2")
    m._max_chunk = 2048
    _check(run_model(kind, m, img.to(cuda), meta.to(cuda)), ref, "f16x2")


def test_empty_batch(cuda):
    kind, cfg = CONFIGS["mm_pico"]
    m = build_model(kind, cfg, seeded_state(kind, cfg, seed=3), cuda, "f32")
    out = run_model(kind, m, torch.zeros(0, 3, 63, 63, device=cuda), torch.zeros(0, 25, device=cuda))
    assert out.shape == (0, 1)


def test_input_validation_and_layout(cuda):
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    m = build_model(kind, cfg, sd, cuda, "f32")
    img, meta, _ = synthetic_batch(4, seed=2)
    with pytest.raises(ValueError):
        m(image_input=img[:, :, :60, :60].to(cuda), metadata_input=meta.to(cuda))
    with pytest.raises(ValueError):
        m(image_input=img.to(cuda), metadata_input=meta[:, :24].to(cuda))
    # a non-contiguous NHWC->NCHW view (what np.transpose gives before ascontiguousarray) and
    # float64 inputs are accepted and give the same answer
    ref = run_model(kind, m, img.to(cuda), meta.to(cuda))
    nhwc = img.permute(0, 2, 3, 1).contiguous().to(cuda)
    out = run_model(kind, m, nhwc.permute(0, 3, 1, 2).double(), meta.to(cuda).double())
    assert torch.equal(out, ref)
    # linearity-free sanity: alerts are independent -> permuting the batch permutes the logits
    perm = torch.tensor([2, 0, 3, 1])
    outp = run_model(kind, m, img[perm].to(cuda), meta[perm].to(cuda))
    assert torch.equal(outp, ref[perm.to(cuda)])


def test_caller_sized_workspace(cuda):
    """btsbot_use_workspace (caller-sized workspace, no allocation inside the library): a forward
    on memory the caller allocated gives the same logits as one on the library's own allocation; a buffer smaller than
    btsbot_workspace_bytes() is refused."""
    import ctypes as C
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(70, seed=2)
    m = build_model(kind, cfg, sd, cuda, "bf16")
    twin = build_model(kind, cfg, sd, cuda, "bf16")
    twin._max_chunk = 32                                   # same chunking on the library's own allocation (an alert's
    ref = run_model(kind, twin, img.to(cuda), meta.to(cuda)).clone()   # logit depends on its position in the chunk
    L = _lib.lib()                                         # through stage1b.hip's summation order, nothing else)
    need = L.btsbot_workspace_bytes(m._handle.ptr, 32)
    assert need > 0
    ws = torch.empty(need + 256, dtype=torch.uint8, device=cuda)
    base = (ws.data_ptr() + 255) // 256 * 256
    assert L.btsbot_use_workspace(m._handle.ptr, 32, C.c_void_p(base), need - 1) != 0     # one byte short
    _lib.check(L.btsbot_use_workspace(m._handle.ptr, 32, C.c_void_p(base), need), "btsbot_use_workspace")
    m._reserved = 32                                       # (the Module's bookkeeping: chunks of 32 are in place)
    m._max_chunk = 32
    out = run_model(kind, m, img.to(cuda), meta.to(cuda))  # 70 alerts = chunks of 32 + 32 + 6 in the caller's memory
    assert torch.equal(out, ref)
    del m                                                  # the library must not free the caller's buffer
    torch.cuda.synchronize()
    ws.fill_(0)


def test_weights_repacked_after_update(cuda):
    kind, cfg = CONFIGS["um_nn"]
    sd = seeded_state(kind, cfg, seed=3)
    m = build_model(kind, cfg, sd, cuda, "f32")
    _, meta, _ = synthetic_batch(16, seed=2)
    a = run_model(kind, m, None, meta.to(cuda))
    with torch.no_grad():
        m.network._modules["6"].bias.add_(1.5)
    b = run_model(kind, m, None, meta.to(cuda))
    assert torch.allclose(b, a + 1.5, atol=1e-4)


def test_large_batch_properties(cuda):
    """BASELINE.json full size (B=1024, bf16): alert independence -- the logits of a big batch
    equal those of its halves run separately, bit for bit (no cross-alert term in eval mode)."""
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    m = build_model(kind, cfg, sd, cuda, "bf16")
    img, meta, _ = synthetic_batch(1024, seed=2)
    img, meta = img.to(cuda), meta.to(cuda)
    full = run_model(kind, m, img, meta)
    lo = run_model(kind, m, img[:512].contiguous(), meta[:512].contiguous())
    hi = run_model(kind, m, img[512:].contiguous(), meta[512:].contiguous())
    assert torch.equal(full, torch.cat([lo, hi]))
    assert torch.isfinite(full).all()


@pytest.mark.parametrize("prec", ["f16x2", "bf16", "f16"])
def test_full_size_batch_matches_oracle(cuda, prec):
    """BASELINE.json configs[1] at its full size (B = 1024, the batch bench.py times) against the oracle -- every
    alert, not a self-comparison.  Stress weights (layer scale gamma ~ 1).  f16x2 is asserted against the north star's
    own 1e-4 (TOL_SCORE)."""
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(1024, seed=2)
    ref = _oracle(kind, cfg, sd, img, meta)
    m = build_model(kind, cfg, sd, cuda, prec)
    ds = _check(run_model(kind, m, img.to(cuda), meta.to(cuda)), ref, prec)
    print(f"B=1024 {prec}: max|dscore| {ds:.3e}")


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_nano_full_size_batch_matches_oracle(cuda, prec):
    """convnext_nano -- the reference classes' default `model_kind` (/root/reference/btsbot/architectures.py:107,128) -- with
    the LS head, at the benchmark's batch (1024 alerts) against the oracle: the per-op schedule (matrix-pipe stem, LDS-staged
    depthwise + LayerNorm, fused MLP for the 80- and 160-channel stages, LDS-DMA GEMMs, the stage-3 and head kernels)."""
    kind, cfg = CONFIGS["mm_nano_ls"]
    sd = seeded_state(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(1024, seed=2)
    ref = _oracle(kind, cfg, sd, img, meta)
    m = build_model(kind, cfg, sd, cuda, prec)
    ds = _check(run_model(kind, m, img.to(cuda), meta.to(cuda)), ref, prec)
    print(f"nano B=1024 {prec}: max|dscore| {ds:.3e}")


@pytest.mark.parametrize("B", [1, 6, 11])
def test_nano_ragged_batches_match_oracle(cuda, B):
    """stage2p at 320 channels keeps 5 alerts per workgroup: a lone alert, one workgroup and an alert, two and an alert
    (the shared row tiles' partial residuals must not leak between the live and the padded pixel columns)."""
    kind, cfg = CONFIGS["mm_nano_ls"]
    sd = seeded_state(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(B, seed=7)
    ref = _oracle(kind, cfg, sd, img, meta)
    for prec in ("bf16", "f16"):
        m = build_model(kind, cfg, sd, cuda, prec)
        _check(run_model(kind, m, img.to(cuda), meta.to(cuda)), ref, prec)


@pytest.mark.timeout(900)
def test_f16x2_meets_the_north_star_on_five_weight_seeds(cuda):
    """The mode that claims the north star's 1e-4 must not owe it to one draw of the weights: five seeded weight sets
    (stress case, layer scale gamma ~ 1) x 1024 alerts each (a fresh synthetic batch per seed), every score against the
    fp32 oracle, asserted against the constant itself (/root/reference/btsbot/to_onnx.py:135-137 is the reference's own
    notion of equal outputs).  The per-seed maxima are printed; VERDICT r3 asks for <= 5e-5 (a 2x margin)."""
    kind, cfg = CONFIGS["mm_pico"]
    worst = []
    for seed in (3, 11, 12, 13, 14):
        sd = seeded_state(kind, cfg, seed=seed)
        img, meta, _ = synthetic_batch(1024, seed=20 + seed)
        ref = _oracle(kind, cfg, sd, img, meta)
        m = build_model(kind, cfg, sd, cuda, "f16x2")
        out = run_model(kind, m, img.to(cuda), meta.to(cuda)).cpu()
        ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs()
        worst.append(ds.max().item())
        print(f"f16x2 weight seed {seed}: max|dscore| {worst[-1]:.3e}, rms {ds.pow(2).mean().sqrt().item():.3e}")
        del m
    assert max(worst) <= NORTH_STAR, worst


@pytest.mark.parametrize("prec", ["bf16", "f16", "fp8"])
def test_seven_alerts_per_workgroup_form_of_stage2(cuda, prec):
    """stage2p.hip keeps 7 alerts (63 of 64 MFMA columns) per workgroup instead of 4 when the batch is large enough
    that this takes fewer rounds of one workgroup per CU (first at 1793 alerts; bench.py's 8192-alert legs run it).
    Same oracle bound as the 4-alert form, on a ragged batch (1795 = 256 x 7 + 3: a partial last workgroup), and
    alert independence against the 4-alert form's logits to the operand-mode's rounding."""
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(256, seed=13)
    reps = 8
    big_img, big_meta = img.repeat(reps, 1, 1, 1)[:1795], meta.repeat(reps, 1)[:1795]
    ref = _oracle(kind, cfg, sd, img, meta).repeat(reps, 1)[:1795]
    m = build_model(kind, cfg, sd, cuda, prec)
    out = run_model(kind, m, big_img.to(cuda), big_meta.to(cuda))
    _check(out, ref, prec)
    small = run_model(kind, m, img.to(cuda), meta.to(cuda))          # 256 alerts: the 4-alert form
    # the two forms add the same products in the same order per alert: identical logits
    assert torch.equal(out[:256], small)


def test_f16_meets_1e4_at_trained_like_layer_scale(cuda):
    """With layer scale ~0.1 (timm initialises it to 1e-6; trained ConvNeXts keep it well below 1) the f16 mode
    is inside the north star's 1e-4 on every alert of a 256-alert batch (measured 5e-5); gamma ~ 1, the
    stress case of the other tests, doubles every block's contribution to the error."""
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3, gamma=0.1)
    img, meta, _ = synthetic_batch(256, seed=2)
    ref = _oracle(kind, cfg, sd, img, meta)
    m = build_model(kind, cfg, sd, cuda, "f16")
    out = run_model(kind, m, img.to(cuda), meta.to(cuda)).cpu()
    ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs().max().item()
    assert ds <= 1e-4, f"f16, gamma 0.1: max|dscore| {ds}"


def test_fp8_mode_error_at_both_layer_scales(cuda):
    """The fp8 mode (stages 2-3 pointwise convolutions on OCP e4m3 operands, one power-of-two scale per filter; the
    rest of the net as in bf16) on 256 alerts: measured max score error 1.8e-2 with layer scale ~1 (the stress case)
    and 1.3e-3 with ~0.1 (trained-like: the size of bf16's own stress-case error); bounds at 2x.  Also: padded and
    ragged batches through the fp8 kernels give finite scores."""
    kind, cfg = CONFIGS["mm_pico"]
    img, meta, _ = synthetic_batch(256, seed=2)
    for gamma, bound in ((1.0, 3.5e-2), (0.1, 2.6e-3)):
        sd = seeded_state(kind, cfg, seed=3, gamma=gamma)
        ref = _oracle(kind, cfg, sd, img, meta)
        m = build_model(kind, cfg, sd, cuda, "fp8")
        out = run_model(kind, m, img.to(cuda), meta.to(cuda)).cpu()
        ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs().max().item()
        assert ds <= bound, f"fp8, gamma {gamma}: max|dscore| {ds}"
        small = run_model(kind, m, img[:5].to(cuda), meta[:5].to(cuda)).cpu()
        assert torch.isfinite(small).all() and (small - out[:5]).abs().max().item() < 1e-3


def test_trained_checkpoint_reproduces_expected_scores(cuda):
    """Auto-activating: the only vector the reference holds for this path is the `expected_scores` column of
    example_data/usage_candidates.csv (inference_example.py:47-95; 8 of its 39 alerts are committed in
    tests/golden/example8.npz).  The published checkpoints cannot be fetched offline; as soon as one is placed
    under models/BTSbot-*/ (pytorch_model.bin + train_config.json, the layout load_HF_model reads), this test
    loads it through the drop-in entry point and requires |score - expected| <= 1e-4 on those alerts for at
    least one of the checkpoints found (the csv does not say which model produced the column)."""
    import glob
    import warnings
    import btsbot_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dirs = [d for d in sorted(glob.glob(os.path.join(root, "models", "BTSbot-*")))
            if os.path.isfile(os.path.join(d, "pytorch_model.bin"))
            and os.path.isfile(os.path.join(d, "train_config.json"))]
    if not dirs:
        pytest.skip("no trained checkpoint under models/BTSbot-*/ (no network here): parity against "
                    "expected_scores stays unpinned")
    ex = np.load(os.path.join(GOLD, "example8.npz"))
    img = torch.from_numpy(ex["triplets"]).to(cuda)
    meta = torch.from_numpy(ex["metadata"]).to(cuda)
    expected = torch.from_numpy(ex["expected_scores"]).float()
    errs = {}
    for d in dirs:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = btsbot_amd.from_HF.load_checkpoint_dir(d, cuda).eval()
        m.set_precision("f32")
        m = m.to(cuda).eval()
        with torch.no_grad():
            try:
                out = m(image_input=img, metadata_input=meta)
            except TypeError:
                out = m(input_data=img)
        errs[os.path.basename(d)] = (torch.sigmoid(out).cpu().reshape(-1) - expected).abs().max().item()
    assert min(errs.values()) <= 1e-4, f"no checkpoint reproduces expected_scores: {errs}"


def test_bce_kernel(cuda):
    g = np.load(os.path.join(GOLD, "adamw_bce.npz"))
    z = torch.from_numpy(g["z"]).to(cuda)
    y = torch.from_numpy(g["y"]).to(cuda)
    loss = torch.zeros(1, device=cuda)
    dz = torch.empty_like(z)
    import ctypes as C
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().btsbot_bce_fwd_bwd(C.c_void_p(z.data_ptr()), C.c_void_p(y.data_ptr()),
                                             float(g["pos_weight"]), z.numel(), z.numel(),
                                             C.c_void_p(loss.data_ptr()), C.c_void_p(dz.data_ptr()),
                                             C.c_void_p(st)), "bce")
    assert abs(loss.item() / z.numel() - float(g["loss"])) < 1e-5
    assert np.allclose(dz.cpu().numpy(), g["dz"], atol=1e-7)


def test_adamw_kernel(cuda):
    g = np.load(os.path.join(GOLD, "adamw_bce.npz"))
    p = torch.from_numpy(g["p0"]).to(cuda)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    import ctypes as C
    st = torch.cuda.current_stream().cuda_stream
    for s in range(3):
        gr = torch.from_numpy(g["grads"][s]).to(cuda)
        _lib.check(_lib.lib().btsbot_adamw_step(
            C.c_void_p(p.data_ptr()), C.c_void_p(gr.data_ptr()), C.c_void_p(m.data_ptr()),
            C.c_void_p(v.data_ptr()), p.numel(), 1e-4, 0.99, 0.99, 1e-8, 1e-2, s + 1,
            C.c_void_p(st)), "adamw")
        assert np.allclose(p.cpu().numpy(), g["traj"][s], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("env", ["BTSBOT_AMD_NO_STAGE2", "BTSBOT_AMD_NO_STAGE0", "BTSBOT_AMD_NO_STAGE1", "BTSBOT_AMD_NO_S3",
                                 "BTSBOT_AMD_NO_HEAD16", "BTSBOT_AMD_NO_STAGE0,BTSBOT_AMD_NO_STEM16"])
@pytest.mark.parametrize("prec", ["bf16", "f16", "f16x2"])
def test_alternative_schedules_match_oracle(cuda, monkeypatch, env, prec):
    """The library's schedule switches (read at model creation) fall back from a stage's fused kernel to
    the per-op launches the other widths (convnext_nano) run -- they must hold the same parity bound.
    (NO_STAGE0 runs the matrix-pipe stem of the per-op schedule, stem16.hip; with NO_STEM16 the fp32 stem_kernel.)"""
    for e in env.split(","):
        monkeypatch.setenv(e, "1")
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(21, seed=5)
    ref = _oracle(kind, cfg, sd, img, meta)
    m = build_model(kind, cfg, sd, cuda, prec)
    _check(run_model(kind, m, img.to(cuda), meta.to(cuda)), ref, prec)


# ---- MaxViT wirings: maxvit_tiny_rw_224 on cutouts resized to 224 ------------
def _mv(name):
    from helpers import MV_CONFIGS, seeded_state_mv
    kind, cfg = MV_CONFIGS[name]
    return kind, cfg, seeded_state_mv(kind, cfg, seed=3)


def _mv_oracle(kind, cfg, sd, img, meta):
    from oracle import maxvit_oracle as MO
    with torch.no_grad():
        return MO.forward(kind, sd, cfg, img, meta)


@pytest.mark.parametrize("name", ["mm_maxvit", "maxvit", "frozen_fusion_maxvit"])
@pytest.mark.parametrize("prec", ["f32", "bf16", "f16"])
def test_maxvit_forward_matches_oracle(cuda, name, prec):
    kind, cfg, sd = _mv(name)
    img, meta, _ = synthetic_batch(5, seed=2)
    ref = _mv_oracle(kind, cfg, sd, img, meta)
    m = build_model(kind, cfg, sd, cuda, prec)
    _check(run_model(kind, m, img.to(cuda), meta.to(cuda)), ref, prec)


@pytest.mark.parametrize("name", ["mm_maxvit", "maxvit", "frozen_fusion_maxvit"])
def test_maxvit_matches_reference_wrapper_goldens(cuda, name):
    kind, cfg, sd = _mv(name)
    gold = np.load(os.path.join(GOLD, "ref_logits_maxvit.npz"))
    ex = np.load(os.path.join(GOLD, "example8.npz"))
    m = build_model(kind, cfg, sd, cuda, "f32")
    img = torch.from_numpy(ex["triplets"][[0, 1, 4, 5]]).to(cuda)
    meta = torch.from_numpy(ex["metadata"][[0, 1, 4, 5]]).to(cuda)
    _check(run_model(kind, m, img, meta), torch.from_numpy(gold[f"{name}/example4"]), "f32")


def test_maxvit_stage_activations_match_oracle(cuda):
    from oracle import maxvit_oracle as MO
    kind, cfg, sd = _mv("mm_maxvit")
    img, meta, _ = synthetic_batch(2, seed=7)
    taps = {}
    with torch.no_grad():
        MO.mm_maxvit_forward(sd, cfg, img, meta, taps=taps)
    m = build_model(kind, cfg, sd, cuda, "f32")
    m.set_debug_taps(True)
    run_model(kind, m, img.to(cuda), meta.to(cuda))
    for t, key in (("stem", "stem"), ("stage0", "s0b1"), ("stage1", "s1b1"), ("stage2", "s2b4"),
                   ("stage3", "s3b1")):
        got = m.read_tap(t).cpu()
        ref = taps[key].permute(0, 2, 3, 1).reshape(got.shape)
        assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), t


def test_maxvit_chunking_independence_and_modes(cuda):
    """Internal workspace chunks (forced to 4 alerts here) and batch permutation do not change a
    logit; the two mixed training regimes of the branch are refused loudly."""
    kind, cfg, sd = _mv("mm_maxvit")
    img, meta, _ = synthetic_batch(10, seed=4)
    img, meta = img.to(cuda), meta.to(cuda)
    m = build_model(kind, cfg, sd, cuda, "bf16")
    full = run_model(kind, m, img, meta)
    m2 = build_model(kind, cfg, sd, cuda, "bf16")
    m2._max_chunk = 4
    assert torch.equal(run_model(kind, m2, img, meta), full)
    perm = torch.randperm(10, generator=torch.Generator().manual_seed(0)).to(cuda)
    assert torch.equal(run_model(kind, m, img[perm].contiguous(), meta[perm].contiguous()), full[perm])
    assert run_model(kind, m, img[:0], meta[:0]).shape == (0, 1)
    # the two mixed cases are refused loudly: a frozen branch in train mode, a trainable branch in eval mode
    # (a trainable branch in train mode trains: tests/test_gpu_train.py::test_maxvit_branch_training_matches_autograd)
    m.train()
    m.maxvit_backbone.eval()                        # eval-mode branch, but its parameters still want gradients
    with pytest.raises(NotImplementedError):
        m(image_input=img, metadata_input=meta)
    for p in m.maxvit_backbone.parameters():
        p.requires_grad_(False)
    m.maxvit_backbone.train()                       # frozen, but BatchNorm2d on batch statistics
    with pytest.raises(NotImplementedError):
        m(image_input=img, metadata_input=meta)


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_maxvit_full_size_batch_is_batch_independent(cuda, prec):
    """BASELINE.json configs[3] size (1100 alerts = an internal chunk of 1024 + one of 76) through a size-independent
    property: an alert's logit does not depend on the batch it came in.  At 1024 alerts per chunk the partition kernels
    walk several units per workgroup (C = 64: persistent workgroups, 128 units each with the next unit's rows in flight;
    C = 128: a grid-stride loop of two), which the oracle-sized batches never reach; the first and last alerts of the big
    batch must equal the same alerts scored on their own, bit for bit (no atomics, no cross-alert arithmetic in the
    forward)."""
    kind, cfg, sd = _mv("mm_maxvit")
    img, meta, _ = synthetic_batch(1100, seed=6)
    img, meta = img.to(cuda), meta.to(cuda)
    m = build_model(kind, cfg, sd, cuda, prec)
    big = run_model(kind, m, img, meta)
    assert torch.isfinite(big).all()
    for sl in (slice(0, 5), slice(1093, 1100)):
        small = run_model(kind, m, img[sl].contiguous(), meta[sl].contiguous())
        assert torch.equal(small, big[sl]), (prec, sl, (small - big[sl]).abs().max().item())


# (the partition blocks of stages 0-2 run as one kernel each by default: the per-op kernels behind them -- LayerNorm fused into a
#  GEMM epilogue, the C = 64 attention block, the register-chained and the streamed MLP -- are reached with NO_PART; those
#  combinations, and the single switches, run in bf16; f16 runs the NO_PART case, which instantiates the same templates)
_MV_SWITCHES = [("BTSBOT_AMD_MV_ATTN_VALU",), ("BTSBOT_AMD_MV_DW_PLAIN",), ("BTSBOT_AMD_MV_STEM_IM2COL",),
                ("BTSBOT_AMD_MV_GATED_GEMM",), ("BTSBOT_AMD_MV_NO_FRONT",), ("BTSBOT_AMD_MV_NO_PART",),
                ("BTSBOT_AMD_MV_NO_PART", "BTSBOT_AMD_MV_MLP_UNFUSED"), ("BTSBOT_AMD_MV_NO_PART", "BTSBOT_AMD_MV_NO_LN_FUSE"),
                ("BTSBOT_AMD_MV_NO_PART", "BTSBOT_AMD_MV_NO_ATTN_BLOCK"), ("BTSBOT_AMD_MV_NO_PART", "BTSBOT_AMD_MV_NO_SMLP")]


_MV_CASES = [(p, e) for p in ("bf16", "f16") for e in _MV_SWITCHES if p == "bf16" or e == ("BTSBOT_AMD_MV_NO_PART",)]


@pytest.mark.parametrize("prec,envs", _MV_CASES,
                         ids=[p + "-" + "+".join(e.replace("BTSBOT_AMD_MV_", "") for e in t) for p, t in _MV_CASES])
def test_maxvit_alternative_kernels_match_oracle(cuda, monkeypatch, envs, prec):
    """16-bit modes default to the partition-block kernels, the MFMA attention kernel (stage 3) and the strip depthwise
    kernel with the fused squeeze-excite pool; the switches select the per-op kernels behind them and the
    one-query-per-lane / per-pixel kernels the f32 mode uses.  All must hold the same bound."""
    for env in envs:
        monkeypatch.setenv(env, "1")
    kind, cfg, sd = _mv("mm_maxvit")
    img, meta, _ = synthetic_batch(3, seed=2)
    ref = _mv_oracle(kind, cfg, sd, img, meta)
    m = build_model(kind, cfg, sd, cuda, prec)
    _check(run_model(kind, m, img.to(cuda), meta.to(cuda)), ref, prec)


def test_score_stream_matches_plain_calls(cuda):
    """btsbot_amd.ScoreStream keeps two / three batches in flight on as many HIP streams (own replica each): same
    logits, in order, as plain model(...) calls, also for ragged batch sizes and a batch count the depth does not divide."""
    import btsbot_amd
    kind, cfg = CONFIGS["mm_pico"]
    m = build_model(kind, cfg, seeded_state(kind, cfg, seed=3), cuda, "bf16")
    batches = []
    for i, n in enumerate((33, 64, 7, 128, 1)):
        img, meta, _ = synthetic_batch(n, seed=20 + i)
        batches.append((img.to(cuda), meta.to(cuda)))
    with torch.no_grad():
        ref = [m(image_input=a, metadata_input=b).clone() for a, b in batches]
    for scorer in (btsbot_amd.ScoreStream(m, depth=2), btsbot_amd.ScoreStream(m)):   # (the default: three in flight)
        for _ in range(2):                               # the second round reuses the streams and replicas
            outs = list(scorer.map(batches))
            torch.cuda.synchronize()
            assert len(outs) == len(ref)
            for o, r in zip(outs, ref):
                assert o.shape == r.shape and torch.equal(o, r)
    with pytest.raises(RuntimeError):
        btsbot_amd.ScoreStream(m.train(), depth=2)
    # a frozen_fusion replica takes its branches from the state dict, not from the checkpoint files
    kind, cfg = CONFIGS["frozen_fusion"]
    f = build_model(kind, cfg, seeded_state(kind, cfg, seed=4), cuda, "f16")
    with torch.no_grad():
        want = f(image_input=batches[1][0], metadata_input=batches[1][1]).clone()
    got = list(btsbot_amd.ScoreStream(f, depth=2).map([batches[1], batches[1], batches[1]]))
    torch.cuda.synchronize()
    assert all(torch.equal(g, want) for g in got)


def test_score_stream_keeps_generator_inputs_alive(cuda):
    """ADVICE r2 (pipeline.py): map() fed from a GENERATOR of freshly allocated batches (what DeviceDataset.__iter__
    and any loader doing .to(device) hand over), with the host `lag` batches ahead of the oldest unfinished one: every
    batch's inputs are dropped by the generator as soon as the next one is pulled, so only the ticket keeps them alive
    while their forward is still queued on a side stream.  Scores must equal plain model(...) calls."""
    import btsbot_amd
    kind, cfg = CONFIGS["mm_pico"]
    m = build_model(kind, cfg, seeded_state(kind, cfg, seed=3), cuda, "bf16")
    img, meta, _ = synthetic_batch(256, seed=31)
    img, meta = img.to(cuda), meta.to(cuda)
    nb = 24
    with torch.no_grad():
        ref = [m(image_input=torch.roll(img, k, 0), metadata_input=torch.roll(meta, k, 0)).clone() for k in range(nb)]

    def gen():
        for k in range(nb):
            a, b = torch.roll(img, k, 0), torch.roll(meta, k, 0)    # new allocations on the caller's stream
            yield a, b
            del a, b
            # what a loader does next: allocate and fill more memory on the caller's stream
            torch.empty_like(img).fill_(float("nan"))

    scorer = btsbot_amd.ScoreStream(m, depth=2)
    outs = []
    for o in scorer.map(gen(), lag=nb + 4):
        outs.append(torch.sigmoid(o))          # a caller-stream consumer of a side-stream allocation
    torch.cuda.synchronize()
    for o, r in zip(outs, ref):
        assert torch.equal(o, torch.sigmoid(r))


def test_config4_fp8_at_its_full_batch(cuda):
    """BASELINE.json configs[4] at its own size: the fp8 mode on 8192 alerts in one call (the library works through it
    in chunks of 7168 + 1024 alerts: the 7-alert and the 4-alert form of stage2p.hip).  The oracle covers a subset --
    the first 256 alerts, to the fp8 mode's bound -- and size-independent properties cover the rest: the batch is 32
    repeats of those 256 alerts, and an alert's logit may depend on its position only through stage1b.hip's chunk
    rotation (a workgroup adds the 16 hidden chunks of fc2 in an order derived from its index, period 256 workgroups =
    512 alerts): every repeat must reproduce the repeat two before it bit for bit -- across workgroups, chunks and the
    two forms of stage2p.hip -- and its neighbour to fp32 summation-order differences; every score is finite and
    inside (0, 1)."""
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(256, seed=2)
    ref = _oracle(kind, cfg, sd, img, meta)
    m = build_model(kind, cfg, sd, cuda, "fp8")
    big_img, big_meta = img.to(cuda).repeat(32, 1, 1, 1), meta.to(cuda).repeat(32, 1)
    out = run_model(kind, m, big_img, big_meta)
    assert out.shape == (8192, 1) and torch.isfinite(out).all()
    _check(out[:256], ref, "fp8")
    for r in range(2, 32):
        assert torch.equal(out[256 * r:256 * (r + 1)], out[256 * (r - 2):256 * (r - 1)]), f"repeat {r} differs from {r - 2}"
    # (a different summation order of the fp32 accumulators moves an fp8 operand across a rounding boundary now and then)
    assert (out[256:512] - out[:256]).abs().max().item() <= 2 * TOL_LOGIT_REL["fp8"] * max(1.0, ref.abs().max().item())
    sc = torch.sigmoid(out)
    assert bool(((sc > 0) & (sc < 1)).all())
