"""CPU: the oracle (oracle/denet_oracle.py) against the fixtures captured
from the reference's own code (oracle/gen_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import denet_oracle as O
from helpers import CASES, load_case, rel_err

TOL = 1e-5    # fp32, same op order as the reference -> essentially identical


@pytest.mark.parametrize("name", CASES)
def test_head_step_matches_reference(name):
    z, d, ep, params, gm, wm = load_case(name)
    gm4 = gm.reshape(-1, d["c"], 1, 1) if gm is not None else None
    out = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"],
                      params, gate_mask=gm4, weight_mask=wm)
    assert rel_err(out["logits_full"], z["out:logits_full"]) < TOL
    assert rel_err(out["logits_binary"], z["out:logits_binary"]) < TOL
    assert rel_err(out["loss_k"], z["out:loss_k"]) < TOL
    assert rel_err(out["loss_p"], z["out:loss_p"]) < TOL
    assert rel_err(out["dFq"], z["out:dFq"]) < 1e-4
    assert rel_err(out["dFs"], z["out:dFs"]) < 1e-4
    for k in params:
        assert rel_err(out["d:" + k], z["grad:" + k]) < 1e-4, k
    if "out:weights" in z.files:
        assert rel_err(out["weights"], z["out:weights"]) < TOL


@pytest.mark.parametrize("name", CASES)
def test_metrics_on_reference_predictions(name):
    z, d, ep, params, gm, wm = load_case(name)
    mat = O.confusion_matrix(ep["Yq_full"].numpy(), z["out:pred_full"].astype(np.int64), d["K"])
    assert np.array_equal(mat, z["out:confmat"])
    miou, _ = O.full_iou_from_mat(mat)
    assert miou == float(z["out:full_miou"])
    bi = O.BinaryIOUOracle()
    pb = z["out:pred_bin"].astype(np.int64)
    if d["way"] > 1:
        pb = pb.reshape(d["N"], d["way"], d["H"], d["W"])
        for i in range(d["way"]):
            bi.update(ep["Yq"][:, i].numpy(), pb[:, i], ep["label"][:, i].numpy())
    else:
        bi.update(ep["Yq"].numpy(), pb, ep["label"].numpy())
    assert list(bi.inter.keys()) == list(z["out:biou_keys"])
    assert np.array_equal(np.array(list(bi.inter.values())), z["out:biou_inter"])
    assert np.array_equal(np.array(list(bi.union.values())), z["out:biou_union"])
    assert bi.mean_iou() == float(z["out:biou_mean"])


def test_metric_boundary_fixture(golden_dir):
    z = np.load(golden_dir + "/metrics.npz")
    K = z["out:mat"].shape[0]
    mat = np.zeros((K, K), dtype=np.int64)
    bi = O.BinaryIOUOracle()
    for u in range(3):
        pred = torch.from_numpy(z[f"in:lp{u}"]).argmax(1).numpy()
        mat += O.confusion_matrix(z[f"in:yt{u}"], pred, K)
        pb = torch.from_numpy(z[f"in:lb{u}"]).max(1)[1].numpy()
        bi.update(z[f"in:yb{u}"], pb, z[f"in:cls{u}"])
    assert np.array_equal(mat, z["out:mat"])
    miou, iu = O.full_iou_from_mat(mat)
    assert miou == float(z["out:miou"])
    assert np.array_equal(iu, z["out:class_iou"])
    assert list(bi.inter.keys()) == list(z["out:biou_keys"])
    assert bi.mean_iou() == float(z["out:biou_mean"])
    assert bi.mean_bf_iou() == float(z["out:biou_bf"])


def test_index_rules(golden_dir):
    z = np.load(golden_dir + "/index_tables.npz")
    for key in z.files:
        kind, a, b = key.split("_")
        a, b = int(a), int(b)
        if kind == "nearest":                       # a = in (H), b = out (h)
            assert np.array_equal(O.nearest_src_index(b, a), z[key]), key
        else:                                       # bilinear a = in (h) -> b = out (H)
            i0, i1, fr = O.bilinear_taps(b, a)
            M = np.zeros((a, b), dtype=np.float32)
            for q in range(b):
                M[i0[q], q] += np.float32(1) - fr[q]
                M[i1[q], q] += fr[q]
            assert np.abs(M - z[key]).max() < 1e-6, key
    assert list(O.nearest_src_index(41, 321)[:6]) == [0, 7, 15, 23, 31, 39]


def test_planted_margin_is_wide(golden_dir):
    z = np.load(golden_dir + "/planted_1way_1shot.npz")
    assert float(z["out:min_margin"]) > 1e-4


def test_trainer_glue_fixture(golden_dir):
    """T1 / N4 rows: up-sampled step outputs and the n-way fusion, oracle vs the reference capture."""
    z = np.load(golden_dir + "/trainer_glue.npz")
    N, way, shot, K, c, h, w, H, W = [int(v) for v in z["shape"]]
    params = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param:")}
    full, binl, _ = O.head_forward(torch.from_numpy(z["in:Fq"]), torch.from_numpy(z["in:Fs"]), torch.from_numpy(z["in:Ys"]),
                                torch.from_numpy(z["in:label"]), params)
    assert rel_err(O.upsample(full, (H, W)), z["out:full_up"]) < TOL
    assert rel_err(O.upsample(binl, (H, W)), z["out:bin_up"]) < TOL
    bi = O.BinaryIOUOracle()
    for u in range(2):
        pre = torch.from_numpy(z[f"in:nway_pre{u}"])
        nway = pre.shape[1] // 2
        fused = O.process_nway(pre, nway)
        assert torch.equal(fused, torch.from_numpy(z[f"out:nway_fused{u}"]))
        Yq, cls = z[f"in:nway_Yq{u}"], z[f"in:nway_cls{u}"]
        for i in range(nway):
            bi.update(Yq[:, i], fused[:, i].max(1)[1].numpy(), cls[:, i])
    assert list(bi.inter.keys()) == z["out:nway_keys"].tolist()
    assert bi.mean_iou() == float(z["out:nway_mean"]) and bi.mean_bf_iou() == float(z["out:nway_bf"])


def test_philox_known_answers_and_dropout_distribution():
    """Random123's published known-answer vectors for philox4x32-10 (kat_vectors: counter, key -> output) pin the
    generator the library's train-mode dropout uses; the multipliers are Bernoulli(1-p) / (1-p)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = tuple(int(x) for x in O.philox4x32_10(*ctr, *key))
        assert got == want
    m = O.dropout_mask(99, 1, 0.5, 1 << 20)
    assert set(np.unique(m).tolist()) == {0.0, 2.0}
    assert abs(float((m != 0).mean()) - 0.5) < 0.01
    # element i depends only on (seed, stream, i): a shorter request is a prefix
    assert np.array_equal(O.dropout_mask(99, 1, 0.5, 1001), m[:1001])
    assert not np.array_equal(O.dropout_mask(99, 2, 0.5, 1001), m[:1001])
    assert np.array_equal(O.dropout_mask(99, 1, 0.0, 7), np.ones(7, np.float32))
