"""Callers either side of the classifier path: on-device dataset + augmentation,
validation metrics, checkpoint hand-over.  CPU tests pin the oracle and the host logic; GPU tests
compare the HIP kernels (through the C ABI) with the oracle -- bit-exact for the index permutations."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

import btsbot_amd
from btsbot_amd import data, to_HF, val
from btsbot_amd.synthetic import synthetic_batch
from helpers import CONFIGS, seeded_state, build_model, run_model
from oracle import data_oracle as DO


# ---- CPU ------------------------------------------------------------------------------------------
def test_oracle_rotation_direction_and_group_structure():
    img = torch.arange(9.0).view(1, 1, 3, 3)
    # 90 degrees counter-clockwise (transforms.functional.rotate's positive direction): the right
    # column becomes the top row
    r1 = DO.augment(img, None, [1 << 2])[0, 0]
    assert r1.tolist() == [[2.0, 5.0, 8.0], [1.0, 4.0, 7.0], [0.0, 3.0, 6.0]]
    x = torch.randn(5, 3, 63, 63, generator=torch.Generator().manual_seed(0))
    assert torch.equal(DO.augment(x, None, [2 << 2] * 5), DO.augment(x, None, [3] * 5))  # rot180 = h.v flips
    twice = DO.augment(DO.augment(x, None, [1 << 2] * 5), None, [3 << 2] * 5)
    assert torch.equal(twice, x)
    idx = torch.tensor([4, 0, 0])
    assert torch.equal(DO.augment(x, idx, [0, 0, 1])[2], torch.flip(x[0], dims=(-1,)))


def test_checkpoint_round_trip(tmp_path, monkeypatch):
    """best_model.pth + report.json -> train_config.json + pytorch_model.bin -> load_HF_model, with a
    DataParallel-style ``module.`` prefix on the way in (to_HF.py:10-43, from_HF.py:59-81)."""
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = btsbot_amd.mm_ConvNeXt(cfg)
    m.load_state_dict(sd)
    mdir = tmp_path / "models" / "BTSbot-convnext-pico-randinit-metadata"
    to_HF.save_checkpoint(m, cfg, str(mdir), {"val_acc": 0.5})
    # a stock-BTSbot DataParallel checkpoint has the prefix: make sure it is accepted
    state = torch.load(mdir / "best_model.pth")
    assert list(state) == list(sd) and all(v.device.type == "cpu" for v in state.values())
    torch.save({"module." + k: v for k, v in state.items()}, mdir / "best_model.pth")
    config = to_HF.prep_config(str(mdir))
    assert config == json.load(open(mdir / "train_config.json")) and config["model_name"] == "mm_ConvNeXt"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        to_HF.prep_model(str(mdir), config)
        monkeypatch.chdir(tmp_path)
        monkeypatch.setattr(btsbot_amd.from_HF, "device", "cpu")
        m2 = btsbot_amd.load_HF_model("convnext", True, "randinit")
    for k, v in m2.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    with pytest.raises(FileNotFoundError):
        to_HF.prep_config(str(tmp_path))


def test_device_dataset_refuses_cpu_and_nans():
    img, meta, lab = synthetic_batch(8, seed=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        data.augment(img, None, None)
    bad = meta.clone()
    bad[0, 0] = float("nan")
    with pytest.raises(ValueError, match="NaNs"):
        data.DeviceDataset(img, bad, lab, 4, device="cpu")
    data.DeviceDataset(None, bad, lab, 4, device="cpu", check_nan=False)     # a validation split is taken as it is
    with pytest.raises(ValueError, match="drop_last"):
        data.DeviceDataset(None, meta, lab, 4, device="cpu", shard=(0, 2), drop_last=False)
    ds = data.DeviceDataset(None, meta, lab, 3, device="cpu")
    assert len(ds) == 2 and abs(ds.pos_weight - ds.num_notbts / max(ds.num_bts, 1)) < 1e-12
    batches = list(ds)                       # metadata-only sets need no kernel
    assert len(batches) == 2 and batches[0][0].shape == (3, 25)


# ---- GPU ------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_augment_kernel_is_the_exact_permutation(cuda):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(37, 3, 63, 63, generator=g)
    idx = torch.randint(0, 37, (64,), generator=g)
    ops = torch.arange(64, dtype=torch.uint8) % 16          # every (hflip, vflip, k) combination
    want = DO.augment(x, idx, ops)
    got = data.augment(x.to(cuda), idx.to(cuda), ops.to(cuda)).cpu()
    assert torch.equal(got, want)
    assert torch.equal(data.augment(x.to(cuda), None, None).cpu(), x)
    assert data.augment(x.to(cuda), idx[:0].to(cuda), ops[:0].to(cuda)).shape == (0, 3, 63, 63)


@pytest.mark.gpu
def test_device_dataset_epoch_is_a_permutation_with_reference_augment_rates(cuda):
    img, meta, lab = synthetic_batch(1000, seed=3)
    meta[:, 0] = torch.arange(1000.0)                      # tag every alert
    gen = torch.Generator(device=cuda).manual_seed(0)
    ds = data.DeviceDataset(img, meta, lab, 64, config={}, device=cuda, generator=gen)
    seen, nflip = [], 0
    for im, me, la in ds:
        assert im.shape == (64, 3, 63, 63) and me.shape == (64, 25) and la.shape == (64,)
        tags = me[:, 0].long().cpu()
        seen.append(tags)
        assert torch.equal(la.cpu(), lab[tags])
        # every augmented cutout is one of the 8 dihedral images of its source
        src = img[tags[0]]
        cands = [torch.rot90(torch.flip(src, dims=(-1,)) if f else src, k, dims=(-2, -1))
                 for f in (0, 1) for k in range(4)]
        assert any(torch.equal(im[0].cpu(), c) for c in cands)
        nflip += int(sum(not torch.equal(im[j].cpu(), img[tags[j]]) for j in range(8)))
    seen = torch.cat(seen)
    assert len(ds) == 15 and seen.numel() == 960 and seen.unique().numel() == 960    # drop_last
    assert nflip > 60                                      # 7/8 of the draws change the image
    ops = ds.draw_ops(20000).cpu()
    assert abs((ops & 1).float().mean().item() - 0.5) < 0.02
    assert abs(((ops >> 1) & 1).float().mean().item() - 0.5) < 0.02
    assert torch.bincount((ops >> 2).long(), minlength=4).min().item() > 4600
    off = data.DeviceDataset(img, meta, lab, 64, device=cuda,
                             config=dict(data_aug_h_flip=False, data_aug_v_flip=False, data_aug_rot=False))
    assert off.draw_ops(4) is None


@pytest.mark.gpu
def test_validation_pass_matches_oracle_metrics(cuda):
    kind, cfg = CONFIGS["mm_pico"]
    sd = seeded_state(kind, cfg, seed=3)
    m = build_model(kind, cfg, sd, cuda, "f32")
    img, meta, lab = synthetic_batch(150, seed=8)
    loss, acc, raw, labels = val.run_val_tensors(m, img, meta, lab, batch_size=64, pos_weight=2.5)
    logits = run_model(kind, m, img.to(cuda), meta.to(cuda)).cpu()
    want_loss, want_acc = DO.metrics(logits, lab, 2.5)
    assert abs(loss - want_loss) < 1e-5 * max(1.0, want_loss) and abs(acc - want_acc) < 1e-6
    assert raw.shape == (150,) and np.allclose(raw, torch.sigmoid(logits).squeeze(1).numpy(), atol=1e-6)
    assert np.array_equal(labels, lab.float().numpy()) and not m.training
    # default pos_weight = N_neg / N_pos of the split (val.py:60-62)
    loss2, _, _, _ = val.run_val_tensors(m, img, meta, lab, batch_size=150)
    pw = float((lab == 0).sum()) / float((lab == 1).sum())
    assert abs(loss2 - DO.metrics(logits, lab, pw)[0]) < 1e-5 * max(1.0, loss2)


@pytest.mark.gpu
def test_train_epoch_over_device_dataset(cuda):
    """train.py:481-566 end to end on the device: augmented batches -> Trainer.step -> epoch metrics."""
    from btsbot_amd.train import Trainer, train_epoch
    kind, cfg = CONFIGS["mm_pico"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = btsbot_amd.mm_ConvNeXt(dict(cfg, meta_dropout=0.0, comb_dropout=0.0), precision="bf16")
    m.load_state_dict(seeded_state(kind, cfg, seed=3))
    m = m.to(cuda).train()
    img, meta, lab = synthetic_batch(192, seed=11)
    ds = data.DeviceDataset(img, meta, lab, 64, config={}, device=cuda,
                            generator=torch.Generator(device=cuda).manual_seed(1))
    tr = Trainer(m, lr=1e-3, betas=(0.9, 0.999), pos_weight=ds.pos_weight, epochs=4, warmup_epochs=1)
    hist = [train_epoch(tr, ds) for _ in range(3)]
    assert all(np.isfinite(l) and 0.0 <= a <= 1.0 for l, a in hist)
    assert tr.epoch == 3 and tr.t == 9


# ---- alert -> triplet arithmetic (alert_utils.py:110-196) -------------------------------------------
def _stamp_cases():
    g = np.random.default_rng(7)
    base = lambda h=63, w=63: np.abs(g.normal(size=(h, w))).astype(np.float32) * 50 + 100
    cases = []
    cases.append([base(), base(), base()])                                  # plain
    a = base(); a[3:9, 10:20] = np.nan; cases.append([a, base(), base()])   # NaN patch
    cases.append([base(63, 40), base(), base(51, 63)])                      # edge stamps: padding
    z = np.zeros((63, 63), np.float32); cases.append([base(), z, base()])   # zero image -> drop
    i = base(); i[:40] = np.inf; cases.append([i, base(), base()])          # inf median -> drop, later stamps raw
    n = np.full((63, 63), np.nan, np.float32); cases.append([base(), base(), n])   # all NaN -> zero -> drop
    m = base(); m[0, 0] = -np.inf; cases.append([m, base(), base()])        # one -inf: norm overflows
    return cases


def test_oracle_prep_matches_example_data_normalisation():
    """The bundled example triplets are what make_triplet produced: every cutout has unit L2 norm."""
    ex = np.load(os.path.join(os.path.dirname(__file__), "golden", "example8.npz"))
    for trip in ex["triplets"][:3]:
        stamps = [trip[c] * 37.0 for c in range(3)]               # undo the scale, redo the arithmetic
        out, drop = DO.make_triplet_arith(stamps)
        assert not drop
        assert np.allclose(np.transpose(out, (2, 0, 1)), trip, rtol=2e-6, atol=1e-9)


@pytest.mark.gpu
def test_prep_triplets_kernel_matches_reference_arithmetic(cuda):
    from btsbot_amd import alert_utils
    cases = _stamp_cases()
    raw, shapes = alert_utils.stack_stamps(cases)
    got, drop = alert_utils.prep_triplets(raw.to(cuda), shapes.to(cuda))
    got, drop = got.cpu().numpy(), drop.cpu().numpy()
    for k, stamps in enumerate(cases):
        want, wdrop = DO.make_triplet_arith(stamps)
        want = np.transpose(want, (2, 0, 1)).astype(np.float32)
        assert bool(drop[k]) == bool(wdrop), k
        both_nan = np.isnan(got[k]) & np.isnan(want)
        assert np.allclose(np.where(both_nan, 0, got[k]), np.where(both_nan, 0, want),
                           rtol=3e-6, atol=1e-12), k
    # all-63x63 fast path (shapes = NULL) and normalize=False
    got2, _ = alert_utils.prep_triplets(raw[:2].to(cuda))
    assert np.array_equal(got2.cpu().numpy(), got[:2])
    raw_only, d = alert_utils.prep_triplets(raw[:1].to(cuda), shapes[:1].to(cuda), normalize=False)
    assert torch.equal(raw_only.cpu(), raw[:1]) and not d.any()


def test_load_split_reads_the_reference_layout(tmp_path):
    """train.py:133-172 / val.py:82-101: file names, NHWC float64 -> NCHW float32, metadata column selection and
    order, NaN triplets dropped from triplets + table + labels in the training split only, NaN metadata refused
    there, uni-modal models skip the file they do not need."""
    import pandas as pd
    from btsbot_amd import data
    from helpers import METADATA_COLS
    rng = np.random.default_rng(3)
    n = 12
    trip = rng.random((n, 63, 63, 3))
    trip[4, 10, 10, 1] = np.nan
    cand = pd.DataFrame({c: rng.random(n) for c in METADATA_COLS[::-1]})      # file order != config order
    cand["label"] = rng.integers(0, 2, n)
    cand["objectId"] = [f"ZTF{i}" for i in range(n)]
    d = tmp_path / "data"
    d.mkdir()
    for split in ("train", "val"):
        np.save(d / f"{split}_triplets_v11_N100.npy", trip)
        cand.to_csv(d / f"{split}_cand_v11_N100.csv", index=False)
    cfg = dict(model_name="mm_ConvNeXt", train_data_version="v11", metadata_cols=METADATA_COLS)
    base = str(tmp_path) + "/"
    t, m, y, c = data.load_split(base, cfg, "train")
    keep = np.arange(n) != 4
    assert t.shape == (n - 1, 3, 63, 63) and t.dtype == torch.float32 and t.is_contiguous()
    assert torch.equal(t, torch.from_numpy(np.transpose(trip[keep].astype(np.float32), (0, 3, 1, 2))))
    assert torch.equal(m, torch.from_numpy(cand[METADATA_COLS].values[keep].astype(np.float32)))
    assert torch.equal(y, torch.from_numpy(cand["label"].values[keep])) and y.dtype == torch.long
    assert list(c["objectId"]) == [f"ZTF{i}" for i in range(n) if i != 4]
    tv, mv, yv, _ = data.load_split(base, cfg, "val")                        # validation keeps every row
    assert tv.shape[0] == n and torch.isnan(tv[4]).any() and yv.shape == (n,)
    t1, m1, _, _ = data.load_split(base, dict(cfg, model_name="um_nn"), "train")
    assert t1 is None and m1.shape == (n, len(METADATA_COLS))                # no triplets read, nothing dropped
    t2, m2, _, _ = data.load_split(base, dict(cfg, model_name="ConvNeXt"), "train")
    assert m2 is None and t2.shape[0] == n - 1
    bad = cand.copy()
    bad.loc[2, METADATA_COLS[3]] = np.nan
    bad.to_csv(d / "train_cand_v11_N100.csv", index=False)
    with pytest.raises(ValueError, match="NaNs found in metadata"):
        data.load_split(base, cfg, "train")
    with pytest.raises(ValueError):
        data.load_split(base, dict(cfg, model_name="nonesuch"), "train")
    with pytest.raises(ValueError, match="Metadata columns"):
        data.load_split(base, dict(model_name="um_nn", train_data_version="v11"), "train")
    with pytest.raises(FileNotFoundError):
        data.load_split(base, dict(cfg, N_max=30), "train")


def test_alert_summary_matches_reference_formulas():
    """val.py:178-218 (np.rint threshold, bitwise confusion masks, sklearn roc_curve + auc) against the torch
    reduction of btsbot_amd.val.alert_summary: heavy ties, a score of exactly 0.5, and the -999 sentinel."""
    from btsbot_amd.val import alert_summary
    rng = np.random.default_rng(1)
    y = (rng.random(5000) < 0.3).astype(np.float32)
    p = np.round(np.clip(0.35 * y + rng.random(5000) * 0.7, 0, 1), 2).astype(np.float32)
    p[:5] = 0.5
    got, want = alert_summary(torch.from_numpy(p), torch.from_numpy(y)), DO.alert_summary(p, y)
    assert set(got) == set(want)
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-12 * max(1.0, abs(want[k])), k
    allneg = alert_summary(torch.full((8,), 0.2), torch.tensor([0, 1, 0, 1, 0, 0, 1, 0]))
    assert allneg["alert_precision"] == -999.0 and allneg["alert_recall"] == -999.0 and allneg["TP"] == 0
    assert allneg["roc_auc"] == 0.5 and allneg["notbts_acc"] == 1.0 and allneg["bts_acc"] == 0.0


def _fits_gz(arr, bitpix=-32, extra=()):
    """An independent writer of the stamp format (gzip of a single-HDU FITS image, 80-character cards in
    2880-byte blocks, big-endian samples, NAXIS1 = fastest axis): what alert packets carry as stampData."""
    import gzip
    cards = [f"{'SIMPLE':<8}= {'T':>20}", f"{'BITPIX':<8}= {bitpix:>20d}", f"{'NAXIS':<8}= {2:>20d}",
             f"{'NAXIS1':<8}= {arr.shape[1]:>20d}", f"{'NAXIS2':<8}= {arr.shape[0]:>20d}",
             f"{'OBJECT':<8}= 'ZTF / cutout'       / a quoted value with a slash"] + list(extra) + ["END"]
    hdr = "".join(c.ljust(80) for c in cards)
    hdr = hdr.ljust((len(hdr) + 2879) // 2880 * 2880)
    data = arr.astype({-32: ">f4", -64: ">f8", 16: ">i2"}[bitpix]).tobytes()
    data += b"\0" * ((-len(data)) % 2880)
    return gzip.compress(hdr.encode("ascii") + data)


def _alert(stamps):
    return {f"cutout{n}": {"stampData": _fits_gz(np.asarray(s, dtype=np.float32))}
            for n, s in zip(("Science", "Template", "Difference"), stamps)}


def test_stamp_decoding_without_astropy():
    """gunzip + FITS primary image (alert_utils.py:139-145): values, NaNs, shape (NAXIS2, NAXIS1), dtype as
    astropy hands them over; scaled integers (BZERO / BSCALE) and float64 images; malformed input raises."""
    from btsbot_amd import alert_utils
    rng = np.random.default_rng(0)
    a = rng.standard_normal((63, 63)).astype(np.float32)
    a[3, 4] = np.nan
    a[10, 0] = np.inf
    out = alert_utils.decode_stamp(_fits_gz(a))
    assert out.dtype == np.float32 and out.shape == (63, 63) and np.array_equal(out, a, equal_nan=True)
    short = rng.standard_normal((61, 50)).astype(np.float32)       # NAXIS2 = 61 rows, NAXIS1 = 50 columns
    assert np.array_equal(alert_utils.decode_stamp(_fits_gz(short)), short)
    i16 = rng.integers(-100, 100, (5, 7)).astype(np.int16)
    sc = alert_utils.decode_stamp(_fits_gz(i16, 16, [f"{'BZERO':<8}= {32768:>20d}", f"{'BSCALE':<8}= {2:>20d}"]))
    assert np.array_equal(sc, i16.astype(np.float32) * 2 + 32768)
    f64 = rng.standard_normal((4, 4))
    assert np.array_equal(alert_utils.decode_stamp(_fits_gz(f64, -64)), f64)
    trip = alert_utils.decode_alert(_alert([a, short, a * 2]))
    assert np.array_equal(trip[1], short) and np.array_equal(trip[2], a * 2, equal_nan=True)
    import gzip
    with pytest.raises(ValueError):
        alert_utils.decode_stamp(gzip.compress(b"not a fits file" * 300))
    with pytest.raises(ValueError):
        alert_utils.decode_fits_image(gzip.decompress(_fits_gz(a))[:2880 + 100])   # truncated data unit


@pytest.mark.gpu
def test_make_triplets_from_alert_packets(cuda):
    """make_triplet end to end (alert_utils.py:110-196): alert packets -> host decode -> one kernel, against the
    reference's own numpy calls on the same stamps (NaN / inf / short / empty cases included)."""
    from btsbot_amd import alert_utils
    cases = _stamp_cases()
    alerts = [_alert(c) for c in cases]
    got, drop = alert_utils.make_triplets(alerts, device=cuda)
    got, drop = got.cpu().numpy(), drop.cpu().numpy()
    assert got.shape == (len(cases), 3, 63, 63) and got.dtype == np.float32
    for k, stamps in enumerate(cases):
        want, wdrop = DO.make_triplet_arith(stamps)
        want = np.transpose(want, (2, 0, 1)).astype(np.float32)
        assert bool(drop[k]) == bool(wdrop), k
        both_nan = np.isnan(got[k]) & np.isnan(want)
        assert np.allclose(np.where(both_nan, 0, got[k]), np.where(both_nan, 0, want), rtol=3e-6, atol=1e-12), k


# ---- examples/inference_example.py (the reference's harness, inference_example.py:47-95) ------------
def _harness():
    import importlib.util
    p = os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "inference_example.py")
    spec = importlib.util.spec_from_file_location("inference_example_amd", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_inference_example_input_preparation():
    """Column list / order, float32 cast and NHWC -> NCHW of the reference harness, on the committed
    example8 fixture (stored as the reference prepares it, so the round trip must be exact)."""
    import pandas as pd
    ex = _harness()
    from btsbot_amd.synthetic import METADATA_COLS
    assert ex.METADATA_COLS == METADATA_COLS and len(ex.METADATA_COLS) == 25
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "example8.npz"))
    cand = pd.DataFrame(g["metadata"], columns=ex.METADATA_COLS)
    cand = cand[ex.METADATA_COLS[::-1]].copy()              # csv column order must not matter
    cand["label"] = g["labels"]
    nhwc = np.transpose(g["triplets"], (0, 2, 3, 1)).astype(np.float64)   # as the .npy file stores them
    img, meta, lab = ex.prepare_inputs(cand, nhwc, True)
    assert img.dtype == torch.float32 and img.is_contiguous() and tuple(img.shape) == (8, 3, 63, 63)
    assert torch.equal(img, torch.from_numpy(g["triplets"]))
    assert torch.equal(meta, torch.from_numpy(g["metadata"])) and lab.dtype == torch.long
    assert ex.prepare_inputs(cand, nhwc, False)[1] is None


@pytest.mark.gpu
def test_inference_example_end_to_end(cuda, tmp_path):
    import pandas as pd
    ex = _harness()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "example8.npz"))
    cand = pd.DataFrame(g["metadata"], columns=ex.METADATA_COLS)
    cand["label"] = g["labels"]
    cand.to_csv(tmp_path / "usage_candidates.csv", index=False)
    np.save(tmp_path / "usage_triplets.npy", np.transpose(g["triplets"], (0, 2, 3, 1)).astype(np.float64))
    kind, cfg = CONFIGS["mm_pico"]
    m = build_model(kind, cfg, seeded_state(kind, cfg, seed=3), cuda, "f32")
    preds, labels = ex.run_inference(m, True, str(tmp_path), device=cuda)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_logits.npz"))["mm_pico/example8"]
    want = torch.sigmoid(torch.from_numpy(gold)).round().squeeze().numpy().astype(int)
    assert np.array_equal(preds, want) and np.array_equal(labels, g["labels"])


@pytest.mark.gpu
def test_fit_loop_checkpoints_and_early_stopping(cuda, tmp_path):
    """train.py:303-352: latest_model.pth every epoch, best_model.pth on a >= 0.5 % better validation loss,
    early stopping; the saved files load strictly into a fresh model."""
    from btsbot_amd.train import Trainer, fit
    kind, cfg = CONFIGS["um_nn"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = btsbot_amd.um_nn(cfg, precision="f32")
    m.load_state_dict(seeded_state(kind, cfg, seed=3))
    m = m.to(cuda).train()
    _, meta, _ = synthetic_batch(512, seed=21)
    lab = (meta[:, 5] > meta[:, 5].median()).long()                     # learnable from one column
    ds = data.DeviceDataset(None, meta[:384], lab[:384], 64, device=cuda,
                            generator=torch.Generator(device=cuda).manual_seed(2))
    tr = Trainer(m, lr=3e-3, betas=(0.9, 0.999), pos_weight=ds.pos_weight, epochs=6, warmup_epochs=1)
    hist = fit(tr, ds, None, meta[384:], lab[384:], str(tmp_path), epochs=6, patience=2, config=cfg)
    n = len(hist["val_loss"])
    assert 1 <= n <= 6 and os.path.isfile(tmp_path / "latest_model.pth") and os.path.isfile(tmp_path / "best_model.pth")
    assert hist["val_loss"][-1] < 0.9 * hist["val_loss"][0] or n < 6      # it learns, or stopped early
    best = torch.load(tmp_path / "best_model.pth")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m2 = btsbot_amd.um_nn(cfg)
    m2.load_state_dict(best, strict=True)
    report = json.load(open(tmp_path / "report.json"))
    assert report["train_config"]["model_name"] == "um_nn"
    assert hist["best_raw_preds"] is not None and hist["best_raw_preds"].shape == (128,)
    # the best epoch's alert-level summary (val.py:178-218) is in the report and equals the reference formulas
    want = DO.alert_summary(hist["best_raw_preds"], hist["best_val_labels"])
    for k, v in want.items():
        assert abs(report["val_summary"][k] - v) <= 1e-9 * max(1.0, abs(v)), k


@pytest.mark.gpu
def test_sharded_device_dataset_is_the_dataparallel_scatter(cuda):
    """shard=(rank, world): with equally seeded generators the ranks draw the same permutation and each yields its
    contiguous slice of every global batch -- concatenated in rank order they ARE the unsharded batch."""
    from btsbot_amd import data
    img, meta, lab = synthetic_batch(200, seed=5)
    mk = lambda **kw: data.DeviceDataset(img, meta, lab, 48, device=cuda, augment=False,
                                         generator=torch.Generator(device=cuda).manual_seed(11), **kw)
    full, r0, r1, r2 = mk(), mk(shard=(0, 3)), mk(shard=(1, 3)), mk(shard=(2, 3))
    assert len(full) == len(r0) == 4
    for epoch in range(2):                                   # the generators advance alike from epoch to epoch
        for bf, b0, b1, b2 in zip(full, r0, r1, r2):
            for k in range(3):
                assert b0[k].shape[0] == 16
                assert torch.equal(bf[k], torch.cat([b0[k], b1[k], b2[k]]))
    with pytest.raises(ValueError):
        mk(shard=(0, 5))                                     # 48 rows do not split five ways
    with pytest.raises(ValueError):
        mk(shard=(3, 3))


@pytest.mark.gpu
def test_run_training_driver_on_split_files(cuda, tmp_path):
    """run_training (train.py:75-440 without WandB / figures) from the reference's split files: um_nn learns a
    column threshold in a few epochs; models/<name>_<version>_N<N>_cuda/<run>/ holds latest / best checkpoints and
    the report with history + val_summary; the best checkpoint loads strictly."""
    import json
    import pandas as pd
    from btsbot_amd.train import run_training
    from helpers import METADATA_COLS
    _, meta, _ = synthetic_batch(768, seed=21)
    lab = (meta[:, 5] > meta[:, 5].median()).long().numpy()
    d = tmp_path / "data"
    d.mkdir()
    for split, sl in (("train", slice(0, 512)), ("val", slice(512, 768))):
        df = pd.DataFrame(meta[sl].numpy(), columns=METADATA_COLS)
        df["label"] = lab[sl]
        df.to_csv(d / f"{split}_cand_v11_N100.csv", index=False)
    cfg = dict(CONFIGS["um_nn"][1], model_name="um_nn", train_data_version="v11", epochs=5, batch_size=64,
               learning_rate="3e-3", warmup_epochs=1, beta_1=0.9, beta_2=0.999, patience=3, random_seed=2)
    hist, model_dir = run_training(cfg, data_base_dir=str(tmp_path) + "/", run_name="t0", device=cuda,
                                   precision="f32", models_root=str(tmp_path / "models"))
    assert model_dir.endswith("um_nn_v11_N100_cuda/t0/") and os.path.isfile(model_dir + "best_model.pth")
    assert os.path.isfile(model_dir + "latest_model.pth")
    n = len(hist["val_loss"])
    assert 1 <= n <= 5 and (hist["val_loss"][-1] < 0.9 * hist["val_loss"][0] or n < 5)
    rep = json.load(open(model_dir + "report.json"))
    assert rep["train_config"]["model_name"] == "um_nn" and len(rep["Training history"]["train_loss"]) == n
    assert 0.0 <= rep["val_summary"]["roc_auc"] <= 1.0 and rep["val_summary"]["TP"] + rep["val_summary"]["FN"] > 0
    import btsbot_amd
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m2 = btsbot_amd.um_nn(cfg)
    m2.load_state_dict(torch.load(model_dir + "best_model.pth"), strict=True)
    # val.py's run_val on the files the run left: the best epoch's validation loss / accuracy again
    from btsbot_amd.val import run_val
    pw = float((lab[:512] == 0).sum() / (lab[:512] == 1).sum())
    vl, va, raw, vlab = run_val(cfg, model_dir, "best_model.pth", pw, data_base_dir=str(tmp_path) + "/",
                                split="val", device=cuda, precision="f32")
    best = int(np.argmin(np.array(hist["val_loss"]) * 1.0))
    assert raw.shape == (256,) and np.array_equal(vlab, lab[512:768].astype(np.float32))
    assert abs(vl - min(hist["val_loss"])) <= 1e-5 * max(1.0, vl) or abs(vl - hist["val_loss"][best]) <= 1e-5
    assert np.allclose(raw, hist["best_raw_preds"], atol=1e-6)
