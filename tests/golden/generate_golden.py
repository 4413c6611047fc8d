#!/usr/bin/env python3
"""Generate the golden vectors in this directory by running the REFERENCE itself.

Runs only in the build container (needs the read-only reference tree at
``/root/reference``); the resulting ``*.npz`` files are committed and are the only thing
that travels.  No reference source is copied: the reference modules are imported from
where they lie and driven through a fake ``worker`` namespace (a ``SimpleNamespace`` with
the attributes ``Attacker`` reads -- attacker.py:24-30,47), which bypasses file I/O only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden.py

Inputs are synthetic and seeded (``linkteller_amd.synth``); every fixture stores its
inputs next to the reference's outputs so tests never need to regenerate anything.
"""
import argparse
import contextlib
import io
import os
import sys
import tempfile
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LT_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from linkteller_amd import synth  # noqa: E402  (our own seeded generators)

# --- reference imports (container only) ------------------------------------------------
import utils as ref_utils          # noqa: E402  /root/reference/utils
from gcn.models import GCN, GCN3   # noqa: E402  /root/reference/gcn/models.py
import attacker as ref_attacker    # noqa: E402  /root/reference/attacker.py
import worker as ref_worker        # noqa: E402  /root/reference/worker.py

torch.set_num_threads(1)  # fixed reduction order inside torch.mm / torch.spmm


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def csr_parts(a):
    a = sp.csr_matrix(a)
    a.sort_indices()
    return dict(indptr=a.indptr.astype(np.int64), indices=a.indices.astype(np.int64),
                data=np.asarray(a.data), n=np.int64(a.shape[0]))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ----------------------------------------------------------------------------------------
# (1) normaliser + scipy->torch conversion       utils/load.py:572-578, 552-559
# ----------------------------------------------------------------------------------------
def gen_normalizer():
    graphs = {
        "er50": synth.erdos_renyi_graph(50, 120, seed=1),
        "pl80": synth.powerlaw_graph(80, 200, seed=2),
    }
    # isolated nodes: drop all edges of nodes 3, 7, 11 from a random graph
    g = synth.erdos_renyi_graph(40, 90, seed=3).tolil()
    for k in (3, 7, 11):
        g[k, :] = 0
        g[:, k] = 0
    g = sp.csr_matrix(g)
    g.eliminate_zeros()
    graphs["iso40"] = g
    # weighted + self loop (graph_reader does not binarise / strip loops, load.py:456-460)
    w = synth.erdos_renyi_graph(30, 60, seed=4).tolil()
    w[2, 2] = 1.0
    w[5, 9] = 2.0
    w[9, 5] = 2.0
    graphs["wl30"] = sp.csr_matrix(w)
    out = {}
    for key, a in graphs.items():
        for norm in ("FirstOrderGCN", "AugNormAdj"):
            res = ref_utils.fetch_normalization(norm)(a)            # scipy COO float64
            t = quiet(ref_utils.sparse_mx_to_torch_sparse_tensor, res)
            out[f"{key}.{norm}.row"] = res.row.astype(np.int64)
            out[f"{key}.{norm}.col"] = res.col.astype(np.int64)
            out[f"{key}.{norm}.data"] = res.data.astype(np.float64)
            out[f"{key}.{norm}.t_indices"] = t._indices().numpy()
            out[f"{key}.{norm}.t_values"] = t._values().numpy()
        for k, v in csr_parts(a).items():
            out[f"{key}.adj.{k}"] = v
    out["keys"] = np.array(sorted(graphs))
    save("normalizer.npz", **out)


# ----------------------------------------------------------------------------------------
# helpers to build a reference model / fake worker
# ----------------------------------------------------------------------------------------
def ref_model(f, h, c, seed, dtype=torch.float32):
    torch.manual_seed(seed)
    m = quiet(GCN, nfeat=f, nhid=h, nclass=c, dropout=0.5)
    m.eval()
    return m.to(dtype)


def state_np(m):
    return {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}


def make_args(**kw):
    base = dict(dataset="twitch/ES/RU", sample_type="unbalanced", n_test=16, sample_seed=42,
                influence=1e-4, mode="vanilla-clean", attack_mode="efficient",
                perturb_type="discrete", epsilon=0.1, noise_seed=42, norm="FirstOrderGCN",
                noise_type="laplace", delta=1e-5)
    base.update(kw)
    return argparse.Namespace(**base)


def fake_worker(adj_clean, adj_served, x, norm, dtype=torch.float32):
    a_hat = ref_utils.fetch_normalization(norm)(adj_served)
    t = quiet(ref_utils.sparse_mx_to_torch_sparse_tensor, a_hat).to(dtype)
    return types.SimpleNamespace(features_2=torch.from_numpy(x).to(dtype), adj_2=t,
                                 features=torch.from_numpy(x).to(dtype), adj_full=t,
                                 adj_ori=sp.csr_matrix(adj_clean), n_nodes=adj_clean.shape[0],
                                 n_features=x.shape[1])


# ----------------------------------------------------------------------------------------
# (2) GCN.forward logits                          gcn/models.py:19-24, gcn/layers.py:30-36
# ----------------------------------------------------------------------------------------
def gen_forward():
    out = {}
    cases = [("n64", 64, 150, 32, 16, 2, False), ("n600", 600, 2400, 128, 256, 2, True),
             ("n200c7", 200, 700, 48, 64, 7, True)]
    for key, n, e, f, h, c, pl in cases:
        a = (synth.powerlaw_graph if pl else synth.erdos_renyi_graph)(n, e, seed=11)
        x = synth.gaussian_features(n, f, seed=12)
        m = ref_model(f, h, c, seed=42)
        a_hat = ref_utils.fetch_normalization("FirstOrderGCN")(a)
        t32 = quiet(ref_utils.sparse_mx_to_torch_sparse_tensor, a_hat)
        with torch.no_grad():
            logits32 = m(torch.from_numpy(x), t32).numpy()
            m64 = ref_model(f, h, c, seed=42, dtype=torch.float64)
            logits64 = m64(torch.from_numpy(x).double(), t32.double()).numpy()
        for k, v in csr_parts(a).items():
            out[f"{key}.adj.{k}"] = v
        out[f"{key}.x"] = x
        for k, v in state_np(m).items():
            out[f"{key}.sd.{k}"] = v
        out[f"{key}.logits32"] = logits32
        out[f"{key}.logits64"] = logits64
    # GCN3 (--n-layer 3), gcn/models.py:28-46
    n, e, f, h1, h2, c = 120, 400, 24, 16, 16, 2
    a = synth.powerlaw_graph(n, e, seed=13)
    x = synth.gaussian_features(n, f, seed=14)
    torch.manual_seed(42)
    m3 = quiet(GCN3, nfeat=f, nhid1=h1, nhid2=h2, nclass=c, dropout=0.5)
    m3.eval()
    t32 = quiet(ref_utils.sparse_mx_to_torch_sparse_tensor, ref_utils.fetch_normalization("FirstOrderGCN")(a))
    with torch.no_grad():
        out["gcn3.logits32"] = m3(torch.from_numpy(x), t32).numpy()
        out["gcn3.logits64"] = m3.double()(torch.from_numpy(x).double(), t32.double()).numpy()
    m3.float()
    for k, v in csr_parts(a).items():
        out[f"gcn3.adj.{k}"] = v
    out["gcn3.x"] = x
    for k, v in state_np(m3).items():
        out[f"gcn3.sd.{k}"] = v
    out["keys"] = np.array([c[0] for c in cases])
    save("forward.npz", **out)


# ----------------------------------------------------------------------------------------
# (3) sampler                                      attacker.py:33-48, utils/load.py:304-381
# ----------------------------------------------------------------------------------------
def gen_sampler():
    a = synth.powerlaw_graph(400, 2400, seed=21)
    out = dict(csr_parts(a))
    out = {f"adj.{k}": v for k, v in out.items()}
    combos = []
    for dataset in ("twitch/ES/RU", "twitch/ES/PTBR"):
        for st in ("unbalanced", "unbalanced-lo", "unbalanced-hi"):
            for seed in (42, 2, 82):
                args = make_args(dataset=dataset, sample_type=st, n_test=24, sample_seed=seed)
                w = types.SimpleNamespace(adj_ori=a, n_nodes=a.shape[0], features_2=None, adj_2=None)
                atk = ref_attacker.Attacker(args, None, w)
                quiet(atk.prepare_test_data)
                tag = f"{dataset.replace('/', '_')}.{st}.{seed}"
                combos.append(tag)
                out[f"{tag}.nodes"] = np.asarray(atk.test_nodes, dtype=np.int64)
                out[f"{tag}.exist"] = np.asarray(atk.exist_edges, dtype=np.int64).reshape(-1, 2)
                out[f"{tag}.nonexist"] = np.asarray(atk.nonexist_edges, dtype=np.int64).reshape(-1, 2)
    out["combos"] = np.array(combos)
    save("sampler.npz", **out)


# ----------------------------------------------------------------------------------------
# (4)+(5) influence matrix, scores, AUC/AP, .pt schema          attacker.py:209-247, 378-412
# ----------------------------------------------------------------------------------------
def run_attack(args, model, w):
    atk = ref_attacker.Attacker(args, model, w)
    quiet(atk.prepare_test_data)
    captured = {}
    orig = atk.compute_and_save

    def spy(norm_exist, norm_nonexist):
        captured["norm_exist"] = np.asarray(norm_exist, dtype=np.float64)
        captured["norm_nonexist"] = np.asarray(norm_nonexist, dtype=np.float64)
        orig(norm_exist, norm_nonexist)

    atk.compute_and_save = spy
    # influence_val is a local in the reference; recover it by replaying its double loop
    # through the reference's own primitive (attacker.py:220-229)
    n = args.n_test
    infl = np.zeros((n, n))
    with torch.no_grad():
        for i in range(n):
            g = atk.get_gradient_eps_mat(atk.test_nodes[i])
            for j in range(n):
                infl[i][j] = g[atk.test_nodes[j]].norm().item()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                atk.link_prediction_attack_efficient()
            files = []
            for root, _, fs in os.walk(td):
                files += [os.path.relpath(os.path.join(root, f), td) for f in fs]
            assert len(files) == 1, files
            saved = torch.load(files[0], weights_only=False)
        finally:
            os.chdir(cwd)
    # the replayed matrix must reproduce the reference's own scores exactly
    node2ind = {node: i for i, node in enumerate(atk.test_nodes)}
    chk = np.array([infl[node2ind[v]][node2ind[u]] for u, v in atk.exist_edges])
    assert np.array_equal(chk, captured["norm_exist"])
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith(("auc =", "ap ="))]
    res = dict(influence_val=infl, test_nodes=np.asarray(atk.test_nodes, dtype=np.int64),
               exist=np.asarray(atk.exist_edges, dtype=np.int64).reshape(-1, 2),
               nonexist=np.asarray(atk.nonexist_edges, dtype=np.int64).reshape(-1, 2),
               norm_exist=captured["norm_exist"], norm_nonexist=captured["norm_nonexist"],
               auc=np.float64(lines[0].split("=")[1]), ap=np.float64(lines[1].split("=")[1]),
               filename=np.array(files[0]),
               fpr=saved["auc"]["fpr"], tpr=saved["auc"]["tpr"], thresholds=saved["auc"]["thresholds"],
               precision=saved["pr"]["precision"], recall=saved["pr"]["recall"],
               pr_thresholds=saved["pr"]["thresholds"],
               y=np.asarray(saved["result"]["y"], dtype=np.int64),
               pred=np.asarray(saved["result"]["pred"], dtype=np.float64),
               schema=np.array(repr({k: sorted(v.keys()) for k, v in saved.items()})),
               y_type=np.array(type(saved["result"]["y"]).__name__),
               pred_type=np.array(type(saved["result"]["pred"][0]).__name__),
               fpr_dtype=np.array(str(saved["auc"]["fpr"].dtype)))
    return res


def gen_influence():
    cases = [
        # key, n, e, f, h, c, powerlaw, n_test, sample_type, mode, perturb
        ("er300", 300, 1500, 64, 32, 2, False, 48, "unbalanced", "vanilla-clean", None),
        ("pl600", 600, 3000, 128, 256, 2, True, 64, "unbalanced", "vanilla-clean", None),
        ("pl600hi", 600, 3000, 128, 256, 2, True, 40, "unbalanced-hi", "vanilla-clean", None),
        ("lap600", 600, 3000, 128, 256, 2, True, 64, "unbalanced", "vanilla", "continuous"),
        ("rand400", 400, 2000, 96, 64, 2, True, 48, "unbalanced", "vanilla", "discrete"),
    ]
    out = {}
    for key, n, e, f, h, c, pl, n_test, st, mode, perturb in cases:
        a = (synth.powerlaw_graph if pl else synth.erdos_renyi_graph)(n, e, seed=31)
        x = synth.twitch_like_features(n, f, seed=32, density=0.05)
        eps = 5.0 if perturb == "continuous" else 4.0
        args = make_args(n_test=n_test, sample_type=st, mode=mode, perturb_type=perturb or "discrete",
                         epsilon=eps)
        served = a
        if perturb is not None:
            # worker.py:632-635 -- the model is served on the perturbed graph, pairs come from
            # the clean adj_ori (worker.py:552)
            fw = types.SimpleNamespace(args=args)
            fn = ref_worker.Worker.perturb_adj_continuous if perturb == "continuous" \
                else ref_worker.Worker.perturb_adj_discrete
            if perturb == "discrete":
                fw.construct_sparse_mat = types.MethodType(ref_worker.Worker.construct_sparse_mat, fw)
            served = sp.csr_matrix(quiet(fn, fw, sp.csr_matrix(a)))
            served.eliminate_zeros()
            for k, v in csr_parts(served).items():
                out[f"{key}.served.{k}"] = v
        m32 = ref_model(f, h, c, seed=42)
        r32 = run_attack(args, m32, fake_worker(a, served, x, args.norm))
        m64 = ref_model(f, h, c, seed=42, dtype=torch.float64)
        r64 = run_attack(args, m64, fake_worker(a, served, x, args.norm, dtype=torch.float64))
        assert np.array_equal(r32["test_nodes"], r64["test_nodes"])
        for k, v in csr_parts(a).items():
            out[f"{key}.adj.{k}"] = v
        out[f"{key}.x"] = x
        for k, v in state_np(m32).items():
            out[f"{key}.sd.{k}"] = v
        for k, v in r32.items():
            out[f"{key}.ref32.{k}"] = v
        for k in ("influence_val", "norm_exist", "norm_nonexist", "auc", "ap"):
            out[f"{key}.ref64.{k}"] = r64[k]
        out[f"{key}.args"] = np.array(repr(vars(args)))
        err = np.abs(r32["influence_val"] - r64["influence_val"]).max()
        print(f"  {key}: max|ref32-ref64| = {err:.3e} on max score {r64['influence_val'].max():.3e};"
              f" auc32={r32['auc']:.6f} auc64={r64['auc']:.6f}")
    out["keys"] = np.array([c[0] for c in cases])
    save("influence.npz", **out)


# ----------------------------------------------------------------------------------------
# (6) DP adjacency generators                                     worker.py:206-335
# ----------------------------------------------------------------------------------------
def gen_dp():
    out = {}
    a = synth.powerlaw_graph(600, 3000, seed=41)
    for k, v in csr_parts(a).items():
        out[f"adj.{k}"] = v
    for perturb, eps in (("continuous", 5.0), ("continuous", 1.0), ("discrete", 4.0), ("discrete", 7.0)):
        args = make_args(perturb_type=perturb, epsilon=eps, noise_seed=42)
        fw = types.SimpleNamespace(args=args)
        fw.construct_sparse_mat = types.MethodType(ref_worker.Worker.construct_sparse_mat, fw)
        fn = ref_worker.Worker.perturb_adj_continuous if perturb == "continuous" \
            else ref_worker.Worker.perturb_adj_discrete
        res = sp.csr_matrix(quiet(fn, fw, sp.csr_matrix(a)))
        res.sort_indices()
        tag = f"{perturb}.eps{eps:g}"
        out[f"{tag}.indptr"] = res.indptr.astype(np.int64)
        out[f"{tag}.indices"] = res.indices.astype(np.int64)
        out[f"{tag}.data"] = np.asarray(res.data, dtype=np.float64)
    save("dp_adjacency.npz", **out)


# ----------------------------------------------------------------------------------------
# (7) "next" rows: baseline attacks, balanced-full, GCN3        attacker.py:250-375, models.py:28-46
# ----------------------------------------------------------------------------------------
def _run_and_capture(atk, method):
    captured = {}
    orig = atk.compute_and_save

    def spy(norm_exist, norm_nonexist):
        captured["norm_exist"] = np.asarray(norm_exist, dtype=np.float64)
        captured["norm_nonexist"] = np.asarray(norm_nonexist, dtype=np.float64)
        orig(norm_exist, norm_nonexist)

    atk.compute_and_save = spy
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                getattr(atk, method)()
            files = []
            for root, _, fs in os.walk(td):
                files += [os.path.relpath(os.path.join(root, f), td) for f in fs]
        finally:
            os.chdir(cwd)
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith(("auc =", "ap ="))]
    captured["auc"] = np.float64(lines[0].split("=")[1])
    captured["ap"] = np.float64(lines[1].split("=")[1])
    captured["filename"] = np.array(files[0])
    return captured


def gen_next_rows():
    out = {}
    n, e, f, h, c = 300, 1400, 64, 32, 2
    a = synth.powerlaw_graph(n, e, seed=51)
    x = synth.twitch_like_features(n, f, seed=52, density=0.05)
    for k, v in csr_parts(a).items():
        out[f"adj.{k}"] = v
    out["x"] = x
    m32 = ref_model(f, h, c, seed=42)
    for k, v in state_np(m32).items():
        out[f"sd.{k}"] = v
    # baseline / baseline-feat on an unbalanced sample (attacker.py:287-334)
    for mode in ("baseline", "baseline-feat"):
        args = make_args(n_test=40, sample_type="unbalanced", attack_mode=mode)
        atk = ref_attacker.Attacker(args, m32, fake_worker(a, a, x, args.norm))
        quiet(atk.prepare_test_data)
        cap = _run_and_capture(atk, "baseline_attack")
        for k, v in cap.items():
            out[f"{mode}.{k}"] = v
        out[f"{mode}.test_nodes"] = np.asarray(atk.test_nodes, dtype=np.int64)
    # balanced-full: sampler + efficient_balanced + baseline_balanced (attacker.py:250-284, 337-375)
    nb_, eb_ = 120, 420
    ab = synth.powerlaw_graph(nb_, eb_, seed=53)
    xb = synth.twitch_like_features(nb_, f, seed=54, density=0.05)
    for k, v in csr_parts(ab).items():
        out[f"bf.adj.{k}"] = v
    out["bf.x"] = xb
    args = make_args(n_test=7, sample_type="balanced-full", attack_mode="efficient", sample_seed=82)
    for dt, tag in ((torch.float32, "ref32"), (torch.float64, "ref64")):
        md = ref_model(f, h, c, seed=42, dtype=dt)
        atk = ref_attacker.Attacker(args, md, fake_worker(ab, ab, xb, args.norm, dtype=dt))
        quiet(atk.prepare_test_data)
        assert args.n_test == nb_
        cap = _run_and_capture(atk, "link_prediction_attack_efficient_balanced")
        for k, v in cap.items():
            out[f"bf.{tag}.{k}"] = v
        if tag == "ref32":
            out["bf.exist"] = np.asarray(atk.exist_edges, dtype=np.int64).reshape(-1, 2)
            out["bf.nonexist"] = np.asarray(atk.nonexist_edges, dtype=np.int64).reshape(-1, 2)
            args_b = make_args(n_test=nb_, sample_type="balanced-full", attack_mode="baseline", sample_seed=82)
            atk_b = ref_attacker.Attacker(args_b, md, fake_worker(ab, ab, xb, args.norm))
            atk_b.exist_edges, atk_b.nonexist_edges, atk_b.test_nodes = atk.exist_edges, atk.nonexist_edges, atk.test_nodes
            capb = _run_and_capture(atk_b, "baseline_attack_balanced")
            for k, v in capb.items():
                out[f"bf.baseline.{k}"] = v
        args.n_test = 7
    # GCN3 served model under the efficient attack (--n-layer 3)
    torch.manual_seed(42)
    m3 = quiet(GCN3, nfeat=f, nhid1=32, nhid2=16, nclass=c, dropout=0.5)
    m3.eval()
    for k, v in state_np(m3).items():
        out[f"gcn3.sd.{k}"] = v
    args3 = make_args(n_test=32, sample_type="unbalanced")
    r32 = run_attack(args3, m3, fake_worker(a, a, x, args3.norm))
    m3d = quiet(GCN3, nfeat=f, nhid1=32, nhid2=16, nclass=c, dropout=0.5)
    m3d.load_state_dict(m3.state_dict())
    m3d.eval()
    r64 = run_attack(args3, m3d.double(), fake_worker(a, a, x, args3.norm, dtype=torch.float64))
    for k in ("influence_val", "test_nodes", "norm_exist", "norm_nonexist", "auc", "ap"):
        out[f"gcn3.ref32.{k}"] = r32[k]
        out[f"gcn3.ref64.{k}"] = r64[k]
    save("next_rows.npz", **out)


# ----------------------------------------------------------------------------------------
# (7) the twitch transfer loader: feature_reader / graph_reader / Worker twitch branch
#     utils/load.py:42-93, 452-460; worker.py:470-496, 549-552, 631-645
# ----------------------------------------------------------------------------------------
def gen_loader():
    """The reference's own ``Worker`` reads a tiny synthetic MUSAE tree (written by
    ``synth.write_musae_dataset``, whose file texts are stored so the test rebuilds exactly these files)
    from ``./data`` of a temporary working directory; everything it exposes to the attack is stored."""
    out = {}
    a1, a2 = synth.powerlaw_graph(220, 500, seed=61), synth.erdos_renyi_graph(200, 450, seed=62)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        synth.write_musae_dataset(os.path.join(td, "data"), "ES", a1, 60, 1)
        synth.write_musae_dataset(os.path.join(td, "data"), "RU", a2, 60, 2)
        for code in ("ES", "RU"):
            for kind in ("features.json", "edges.csv", "target.csv"):
                with open(os.path.join(td, "data", "twitch", code, f"musae_{code}_{kind}")) as fh:
                    out[f"file.{code}.{kind}"] = np.array(fh.read())
        os.chdir(td)
        cuda_was = torch.cuda.is_available
        torch.cuda.is_available = lambda: False
        try:
            for mode, tag in (("vanilla-clean", "clean"), ("vanilla", "lap5")):
                args = argparse.Namespace(mode=mode, norm="FirstOrderGCN", perturb_type="continuous", epsilon=5.0,
                                          noise_seed=42, noise_type="laplace", delta=1e-5)
                w = quiet(ref_worker.Worker, args, dataset="twitch/ES/RU", mode=mode)
                if tag == "clean":
                    out["features_1"] = w.features_1.numpy()
                    out["features_2"] = w.features_2.numpy()
                    out["labels_1"] = w.labels_1.numpy()
                    out["labels_2"] = w.labels_2.numpy()
                    out["sizes"] = np.array([w.n_nodes_1, w.n_nodes_2, w.n_features, w.n_classes, w.multi_label],
                                            dtype=np.int64)
                    for k, v in csr_parts(w.adj_ori).items():
                        out[f"adj_ori.{k}"] = v
                    out["adj_ori.dtype"] = np.array(str(w.adj_ori.dtype))
                for name in ("adj_1", "adj_2"):
                    t = getattr(w, name).coalesce()
                    out[f"{tag}.{name}.indices"] = t.indices().numpy()
                    out[f"{tag}.{name}.values"] = t.values().numpy()
                    out[f"{tag}.{name}.shape"] = np.array(t.shape, dtype=np.int64)
                # the clean graph stays the ground truth for the pairs (worker.py:552)
                assert (w.adj_ori != (a2 + 0)).nnz == 0
        finally:
            torch.cuda.is_available = cuda_was
            os.chdir(cwd)
    save("twitch_loader.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["normalizer", "forward", "sampler", "influence", "dp", "next", "loader"]
    for w in which:
        print(f"[{w}]")
        {"normalizer": gen_normalizer, "forward": gen_forward, "sampler": gen_sampler,
         "influence": gen_influence, "dp": gen_dp, "next": gen_next_rows, "loader": gen_loader}[w]()
