"""GPU, 2 processes: a two-layer fit with one layer per process (layer-sharded mode, 3-double all-reduce per
sweep over gloo) reproduces the reference's joint two-layer fit (golden case B: K = 3, random mask, mutuality)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from tests.golden_util import case_config, load_case
    from vimure_amd.sharded import fit_layer_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = load_case("B_random_mask_K3")
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    res = fit_layer_sharded(d["X"][rank:rank + 1], d["R"][rank:rank + 1], [rank], 2, K, dist, seed=seed, mutuality=mut,
                            device=0, **fitargs)
    q.put((rank, [t[2] for t in res["trace"]], [t[3] for t in res["trace"]], [t[1] for t in res["trace"]],
           res["maxL"], res["posterior"]["rho"], res["posterior"]["nu_shp"], res["next_seed"]))
    dist.destroy_process_group()


def test_two_ranks_one_layer_each_equal_the_joint_fit():
    import torch.multiprocessing as mp
    from tests.golden_util import load_case
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = load_case("B_random_mask_K3")
    for rank, iters, elbos, seeds, maxL, rho, nu_shp, next_seed in got:
        assert iters == d["fit_trace_iter"].tolist() and seeds == d["fit_trace_seed"].tolist()
        ref = d["fit_trace_elbo"]
        assert np.all(np.abs(np.array(elbos) - ref) <= 1e-8 * np.maximum(1.0, np.abs(ref)))
        assert abs(maxL - float(d["fit_maxL"])) <= 1e-8 * abs(float(d["fit_maxL"]))
        np.testing.assert_allclose(rho[0], d["fit_rho_f"][rank], rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(nu_shp, d["fit_nu_shp_f"], rtol=1e-7)
        assert next_seed == int(d["fit_final_seed"])


def _worker_batch(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from tests.golden_util import load_case
    from vimure_amd.batch import fit_datasets
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = load_case("D_self_mask")
    data = {"a": (d["X"], d["R"]), "b": (d["X"][:1], d["R"][:1]), "c": (d["X"][1:], d["R"][1:])}
    df = fit_datasets(data, K=2, seeds=[1, 2], dist=dist, device=0, workers=2, num_realisations=2, max_iter=21)
    q.put((rank, df[["dataset", "layer", "seed", "iters"]].values.tolist(), df["elbo"].tolist()))
    dist.destroy_process_group()


def test_fit_datasets_sharded_over_two_ranks_equals_one_process():
    """The (dataset, layer) units of the batch driver sharded over two ranks (one process per GPU in production; both on
    this box's GPU here), rows gathered with one all_gather: every rank ends with the table one process computes."""
    import torch.multiprocessing as mp
    from tests.golden_util import load_case
    from vimure_amd.batch import fit_datasets
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_batch, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = load_case("D_self_mask")
    data = {"a": (d["X"], d["R"]), "b": (d["X"][:1], d["R"][:1]), "c": (d["X"][1:], d["R"][1:])}
    one = fit_datasets(data, K=2, seeds=[1, 2], workers=1, num_realisations=2, max_iter=21)
    for rank, keys, elbos in got:
        assert keys == one[["dataset", "layer", "seed", "iters"]].values.tolist()
        np.testing.assert_allclose(elbos, one["elbo"].values, rtol=1e-9)


def test_device_resident_nu_exchange_equals_the_fused_sweep():
    """vmr_sweep_local_dev / vmr_commit_nu_dev / vmr_stream (the exchange of a layer-sharded fit without a host hop): with ONE
    owner the all-reduce is the identity, so sweeps driven through the device buffer -- queued on the engine's own stream
    (torch.cuda.ExternalStream) -- must leave exactly the state and ELBO of `vmr_step`."""
    import scipy.special as sp
    import torch
    from oracle import vimure_oracle as vo
    from tests.golden_util import case_config, load_case
    from vimure_amd import CaviEngine
    d = load_case("B_random_mask_K3")
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    L, N, _, M = d["X"].shape
    pr = vo.make_priors(L, M, K, **priors)
    pb = vo.Problem(d["X"], d["R"], K, mut, pr, undirected=und)
    st = vo.init_state(pb, np.random.RandomState(seed), rho_prior=rho_prior)
    out = []
    for dev_path in (False, True):
        eng = CaviEngine(d["X"], d["R"], K=K, mutuality=mut)
        eng.set_priors(pr.alpha_theta, pr.beta_theta, pr.alpha_lambda, pr.beta_lambda, pr.alpha_eta, pr.beta_eta)
        eng.set_state(st.gamma_shp, st.gamma_rte, st.phi_shp, st.phi_rte, st.nu_shp, st.nu_rte, st.pr_rho)
        if not dev_path:
            e = eng.step(3, want_elbo=True)
        else:
            assert eng.stream_ptr() != 0
            ext = torch.cuda.ExternalStream(eng.stream_ptr(), device=torch.device("cuda:0"))
            buf = torch.zeros(3, dtype=torch.float64, device="cuda:0")
            for it in range(3):
                eng.sweep_local_dev(buf, want_elbo=(it == 2))
                with torch.cuda.stream(ext):
                    buf.mul_(1.0)            # where the all-reduce of several owners sits: on the engine's stream
                eng.commit_nu_dev(buf)
            with torch.cuda.stream(ext):
                tot = buf.cpu().tolist()
            g = eng.get_state(rho=False)
            assert abs(g["nu_shp"] - (pr.alpha_eta + tot[0])) <= 1e-12 * abs(g["nu_shp"])
            e = tot[1] - (g["nu_shp"] / g["nu_rte"]) * tot[2] + float(
                sp.gammaln(g["nu_shp"]) - pr.alpha_eta * np.log(g["nu_rte"]) + (pr.alpha_eta - g["nu_shp"]) * sp.psi(g["nu_shp"])
                + g["nu_shp"] * (1.0 - pr.beta_eta / g["nu_rte"]))
        out.append((e, eng.get_state(rho=True)))
        eng.close()
    (e0, s0), (e1, s1) = out
    assert abs(e0 - e1) <= 1e-10 * abs(e0), (e0, e1)
    for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "rho"):
        np.testing.assert_allclose(s1[k], s0[k], rtol=1e-11, atol=1e-13, err_msg=k)


def _worker_nccl(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    import torch
    import torch.distributed as dist
    from tests.golden_util import case_config, load_case
    from vimure_amd.sharded import fit_layer_sharded
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    d = load_case("B_random_mask_K3")
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    res = fit_layer_sharded(d["X"][rank:rank + 1], d["R"][rank:rank + 1], [rank], 2, K, dist, seed=seed, mutuality=mut,
                            device=rank, **fitargs)
    q.put((rank, [t[3] for t in res["trace"]], res["maxL"], res["posterior"]["nu_shp"], dist.get_world_size()))
    dist.destroy_process_group()


def test_two_ranks_over_rccl_when_two_gpus_are_visible():
    """The same joint fit with one layer per GPU over RCCL (backend "nccl"): the 3-double exchange stays on the devices.
    Needs two GPUs; the one-GPU boxes of this project skip it (the gloo test above covers the protocol)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    from tests.golden_util import load_case
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_nccl, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = load_case("B_random_mask_K3")
    for rank, elbos, maxL, nu_shp, ws in got:
        assert ws == 2
        ref = d["fit_trace_elbo"]
        assert np.all(np.abs(np.array(elbos) - ref) <= 1e-8 * np.maximum(1.0, np.abs(ref)))
        assert abs(maxL - float(d["fit_maxL"])) <= 1e-8 * abs(float(d["fit_maxL"]))
        np.testing.assert_allclose(nu_shp, d["fit_nu_shp_f"], rtol=1e-7)


class _DeviceTensorsOverGloo:
    """A stand-in for torch.distributed with ONE GPU shared by two ranks: it reports a backend other than gloo, so
    `fit_layer_sharded` takes its device-resident branch (vmr_sweep_local_dev -> all_reduce on the engine's stream ->
    vmr_commit_nu_dev), and carries the all-reduce of the CUDA tensor through the host over gloo.  Everything of the two-rank
    exchange but RCCL itself."""

    def __init__(self, dist):
        self._d = dist

    def get_backend(self):
        return "nccl-stand-in"

    def all_reduce(self, t, op=None):
        c = t.detach().cpu()      # (on the current stream: inside `with torch.cuda.stream(ext)` that is the engine's own)
        self._d.all_reduce(c)
        t.copy_(c)

    def __getattr__(self, name):
        return getattr(self._d, name)


def _worker_dev(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from tests.golden_util import case_config, load_case
    from vimure_amd.sharded import fit_layer_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = load_case("B_random_mask_K3")
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    res = fit_layer_sharded(d["X"][rank:rank + 1], d["R"][rank:rank + 1], [rank], 2, K, _DeviceTensorsOverGloo(dist), seed=seed,
                            mutuality=mut, device=0, **fitargs)
    q.put((rank, [t[2] for t in res["trace"]], [t[3] for t in res["trace"]], res["maxL"], res["posterior"]["nu_shp"]))
    dist.destroy_process_group()


def test_two_ranks_device_resident_exchange_on_one_gpu():
    """The device-resident exchange (the branch RCCL takes) with TWO ranks, on the one GPU of this box: the reference's joint
    two-layer fit again (golden case B)."""
    import torch.multiprocessing as mp
    from tests.golden_util import load_case
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_dev, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = load_case("B_random_mask_K3")
    for rank, iters, elbos, maxL, nu_shp in got:
        assert iters == d["fit_trace_iter"].tolist()
        ref = d["fit_trace_elbo"]
        assert np.all(np.abs(np.array(elbos) - ref) <= 1e-8 * np.maximum(1.0, np.abs(ref)))
        assert abs(maxL - float(d["fit_maxL"])) <= 1e-8 * abs(float(d["fit_maxL"]))
        np.testing.assert_allclose(nu_shp, d["fit_nu_shp_f"], rtol=1e-7)
