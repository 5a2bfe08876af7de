"""Data-parallel path on CPU: 2 ranks over gloo (127.0.0.1), each running the product's
SoftIntroStep on its shard with the test-only CPU backend, against a single-process emulation
(shards run one after the other, gradients averaged by hand, BatchNorm statistics per shard)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG = (4, [[4, 1, 2], [8, 1, 2], [8, 2, 2]])


def _setup_paths():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "soft-intro-vae-for-3d-mri_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _load_sd():
    g = np.load(os.path.join(GOLDEN, "g4_step.npz"), allow_pickle=False)
    sd = {k[4:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd0.")}
    real = torch.cat([torch.from_numpy(g["s0.real"]), torch.from_numpy(g["s1.real"])])[:4]
    noise = torch.cat([torch.from_numpy(g["s0.noise"]), torch.from_numpy(g["s1.noise"])])[:4]
    return sd, real, noise, float(g["hp.scale"])


def _one_rank_step(sd, real, noise, scale, rank, sync):
    """Returns (E grads after sync, D grads after sync, params after the step)."""
    from oracle import siv_oracle as O
    import siv.rng as srng
    import siv.models.models as models
    import siv.utils.my_trainer as trainer
    from siv.optim import FlatAdam
    net = models.SoftIntroVAE(*CFG)
    net.load_state_dict(sd)
    net.train()
    oe, od = FlatAdam(net.encoder.parameters(), lr=2e-4), FlatAdam(net.decoder.parameters(), lr=2e-4)
    step = trainer.SoftIntroStep(net, oe, od, 1.0, 1024.0, 0.75, 1e-8, scale, sync)
    cap = {}
    se, sdd = oe.step, od.step
    oe.step = lambda: (cap.__setitem__("gE", oe.flat_g.clone()), se())
    od.step = lambda: (cap.__setitem__("gD", od.flat_g.clone()), sdd())
    with srng.use_rng(srng.ReplayRng(O.Draws(seed=100 + rank))):
        step(real, noise)
    return cap["gE"], cap["gD"], oe.flat_p.clone(), od.flat_p.clone()


def _worker(rank, world, port, out):
    _setup_paths()
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import siv.backend as sb
    from tests.cpu_backend import CpuCheckerBackend
    from siv.parallel import GradSync, init_process_group
    sb.set_backend(CpuCheckerBackend())
    init_process_group("gloo")
    sync = GradSync(None)
    assert sync.world == world and sync.rank == rank
    sd, real, noise, scale = _load_sd()
    res = _one_rank_step(sd, sync.shard(real), sync.shard(noise), scale, rank, sync)
    out[rank] = [t.numpy() for t in res]
    # initial broadcast: ranks that were initialised differently end up with rank 0's replica
    import siv.models.models as models
    torch.manual_seed(1000 + rank)
    net = models.SoftIntroVAE(*CFG)
    with torch.no_grad():
        for b in net.buffers():
            if b.dtype == torch.int64:
                b.fill_(5 + rank)
    sync.broadcast_(net)
    out[f"bcast{rank}"] = {k: v.numpy().copy() for k, v in net.state_dict().items()}
    # the trainers shard the loader's (global) batches themselves and log global-batch means
    import siv.rng as srng
    import siv.utils.my_trainer as trainer
    from oracle import siv_oracle as O
    import tempfile
    gen = torch.Generator().manual_seed(7)
    loader = [(torch.rand(4, 1, 16, 24, 16, generator=gen), torch.zeros(4)) for _ in range(2)]
    torch.manual_seed(2000 + rank)                                       # different start on every rank
    net = models.SoftIntroVAE(*CFG)
    with tempfile.TemporaryDirectory() as d, srng.use_rng(srng.ReplayRng(O.Draws(seed=300 + rank))):
        lists = trainer.train_soft_intro_vae(net, loader, loader[:1], 1, 2e-4, torch.device("cpu"), d + "/",
                                             1.0, 1024.0, 0.75)
        files = sorted(os.listdir(d))
    out[f"train{rank}"] = (np.array(lists[0]), np.array(lists[1]),
                           {k: v.numpy().copy() for k, v in net.state_dict().items() if "running" not in k
                            and "num_batches" not in k}, files)
    dist.barrier()
    dist.destroy_process_group()


class _NoSync:
    world, rank = 1, 0
    pending = False

    def finish_step(self, opt, stats):
        self.average_and_step(opt)
        return stats

    def average_(self, flat_g):
        pass

    def average_and_step(self, opt):
        opt.step()

    def wait(self):
        pass

    def average_stats_(self, t):
        return t


class _InjectSync(_NoSync):
    """Sequential emulation of the exchange: records this shard's gradient of the n-th optimiser step and
    replaces it with the given all-shard average (None = keep)."""

    def __init__(self, inject):
        self.inject, self.seen = list(inject), []

    def average_and_step(self, opt):
        self.seen.append(opt.flat_g.clone())
        avg = self.inject[len(self.seen) - 1]
        if avg is not None:
            opt.flat_g.copy_(avg)
        opt.step()


def test_two_rank_gloo_matches_sequential_shards():
    _setup_paths()
    import siv.backend as sb
    from tests.cpu_backend import CpuCheckerBackend
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    # both ranks hold the same averaged gradients and the same weights after the step
    for a, b in zip(r0, r1):
        np.testing.assert_array_equal(a, b)
    # emulate the exchange by hand, shard after shard: encoder gradients first (they do not depend on the
    # exchange), then -- with the averaged encoder gradient injected, so that the D-step runs on the same
    # updated encoder as in the two-rank job -- the decoder gradients, then the final weights
    sb.set_backend(CpuCheckerBackend())
    try:
        sd, real, noise, scale = _load_sd()
        shard = lambda r: (real[2 * r:2 * r + 2], noise[2 * r:2 * r + 2])
        gE = [_one_rank_step(sd, *shard(r), scale, r, _NoSync())[0] for r in range(2)]
        avgE = (gE[0] + gE[1]) / 2
        gD = []
        for r in range(2):
            sy = _InjectSync([avgE, None])
            _one_rank_step(sd, *shard(r), scale, r, sy)
            gD.append(sy.seen[1])
        avgD = (gD[0] + gD[1]) / 2
        final = _one_rank_step(sd, *shard(0), scale, 0, _InjectSync([avgE, avgD]))
    finally:
        sb.set_backend(None)
    tol = lambda a: dict(rtol=1e-5, atol=1e-5 * float(np.abs(a).max()))
    np.testing.assert_allclose(r0[0], avgE.numpy(), **tol(r0[0]))                 # encoder gradient buffer
    np.testing.assert_allclose(r0[1], avgD.numpy(), **tol(r0[1]))                 # decoder gradient buffer
    # weights after the step: Adam's first update is lr * sign(g) -- tight where |g| is clear of rounding noise,
    # +-2 lr elsewhere (the two-rank sum and the emulation add the shard gradients in different orders)
    from tests.parity import check_params_after_adam
    check_params_after_adam(r0[2], final[2], avgE, 2e-4, "encoder weights after the step")
    check_params_after_adam(r0[3], final[3], avgD, 2e-4, "decoder weights after the step")

    # broadcast: rank 1 (seeded differently) now holds rank 0's parameters and buffers
    b0, b1 = out["bcast0"], out["bcast1"]
    torch.manual_seed(1000)
    import siv.models.models as models
    want = models.SoftIntroVAE(*CFG).state_dict()
    for k in b0:
        np.testing.assert_array_equal(b0[k], b1[k], err_msg=k)
        if b0[k].dtype == np.int64:
            assert int(b0[k]) == 5, k
        else:
            np.testing.assert_array_equal(b0[k], want[k].numpy(), err_msg=k)

    # trainer: ranks started from different weights, trained on different halves of each batch, and end with
    # identical weights and identical (global-mean) loss lists; only rank 0 wrote checkpoints and logs
    t0, t1 = out["train0"], out["train1"]
    np.testing.assert_array_equal(t0[0], t1[0])
    np.testing.assert_array_equal(t0[1], t1[1])
    assert np.isfinite(t0[0]).all() and np.isfinite(t0[1]).all()
    for k in t0[2]:
        np.testing.assert_array_equal(t0[2][k], t1[2][k], err_msg=k)
    assert "loss.txt" in t0[3] and "prams" in t0[3] and "loss.txt" not in t1[3]


def test_shard_rejects_indivisible_batch():
    _setup_paths()
    from siv.parallel import GradSync
    g = GradSync(None)
    g.world, g.rank = 3, 1
    try:
        g.shard(torch.zeros(4, 1))
        raise AssertionError("expected ValueError")
    except ValueError:
        pass
    g.world = 2
    assert g.shard(torch.arange(8).view(4, 2)).tolist() == [[4, 5], [6, 7]]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (no RANK / WORLD_SIZE in the environment) must start two
    ranks itself and rank 0 must print the one JSON line for the two-rank job.  Run here over gloo with the
    test-only checker backend injected (`--rehearse-cpu`): the launcher, the rendezvous, the sharded step with
    its two gradient exchanges and the max-over-ranks timing are the code the GPU job runs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SIV_FORCE_COLLECTIVE")}
    env["SIV_BENCH_BACKEND"] = "tests.cpu_backend:CpuCheckerBackend"
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--rehearse-cpu"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # exactly one line on stdout, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["global_batch"] == 2 * out["config"]["batch_per_gpu"]
    assert out["steps"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    assert all(np.isfinite(v) for v in out["losses"].values())

    # a rank that fails takes the job down with a non-zero exit code instead of hanging the others
    env_bad = dict(env)
    del env_bad["SIV_BENCH_BACKEND"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--rehearse-cpu"], env=env_bad, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
