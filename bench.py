#!/usr/bin/env python3
"""Benchmark of the Soft-IntroVAE training step (E-step + D-step, both Adam updates) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no RANK / WORLD_SIZE in the environment) it is its own launcher: the
parent starts N fresh child processes, one per GPU, before it has imported torch or touched a GPU, and only
waits for them (it never becomes a rank and never replaces itself); rank 0's child prints the JSON line.
Under torch.distributed.run the ranks already exist and each process is one of them.

A "step" is one full iteration of utils/my_trainer.py:242-325 on one batch of synthetic volumes
that is already resident in HBM.  Workload: BASELINE.json configs[2] -- 80x96x80 fp32, z=1200,
SoftIntroVAE(64,[[64,1,2],[128,1,2],[256,2,2]]), batch 8 per GPU (weak scaling: global batch
8*N, rank gradients averaged with RCCL all-reduce before each optimiser step).

Rank 0 prints ONE JSON line: volumes/s over all GPUs, the roofline of the dominant kernel
(measured live with HIP events on the launch stream inside the timed region) and, at N=1, the
CPU oracle timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="volumes per GPU (default 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fc", action="store_true", help="skip the FC-latent z=600 side measurement")
    ap.add_argument("--graph", action="store_true",
                    help="single GPU: capture the iteration in a HIP graph and replay it (the per-kernel "
                         "event timing of the roofline fields is then taken from two eager iterations)")
    ap.add_argument("--no-f2x4", action="store_true",
                    help="skip the separately reported run with the opt-in F(2,3) x F(4,3) Winograd kernel")
    ap.add_argument("--split-bf16", action="store_true",
                    help="also time the opt-in conv path 5 (fp32 products from bf16 triples); never the headline")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="NOT a measurement: run the launcher and the rank code on CPU over gloo with a narrow "
                         "model on 16x24x16 volumes; the arithmetic backend is the class named by "
                         "SIV_BENCH_BACKEND=module:Class (the test suite passes its checker), there is no "
                         "built-in CPU path")
    return ap.parse_args(argv)


def launch_ranks(n: int, argv) -> int:
    """Start ``n`` ranks of this script as fresh child processes (RANK / LOCAL_RANK = 0..n-1, one rendezvous
    on 127.0.0.1) and wait for them.  Called before torch is imported: the parent holds no GPU state, does
    not join the job and does not replace itself.  Returns the first non-zero exit code (the remaining
    ranks are then stopped), else 0.  stdout is inherited, so rank 0's JSON line is the parent's output."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver supports dmabuf IPC only
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            time.sleep(0.2)
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    if code != 0:
                        rc = code
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


if __name__ == "__main__":
    _a = parse_args()
    if _a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(_a.gpus, sys.argv[1:]))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MODEL = (64, [[64, 1, 2], [128, 1, 2], [256, 2, 2]])     # z-1200main.py:158
DIMS = (80, 96, 80)
PER_GPU_BATCH = 8
REHEARSAL_MODEL = (4, [[4, 1, 2], [8, 1, 2], [8, 2, 2]])     # --rehearse-cpu only
REHEARSAL_DIMS = (16, 24, 16)
FP32_PEAK_TFLOPS = 157.3                                   # MI355X fp32 matrix/vector peak
HBM_PEAK_TBS = 8.0                                         # MI355X HBM3E peak (about 6.3 TB/s reachable by a copy)
# dominant kernel: 3x3x3 conv 64->64 at 80x96x80 (60 % of the model's FLOPs):
# 2 * 27 * 64 * 64 * 614400 = 135.9 GFLOP per volume per launch
DOM_FLOP_PER_VOLUME = 2.0 * 27 * 64 * 64 * DIMS[0] * DIMS[1] * DIMS[2]
WINO2D_ISSUED = 12.0 / 27.0        # Winograd F(2x2,3x3) over (H,W): multiply-adds issued per direct one
WGRAD_ISSUED = 16.0 / 36.0         # Winograd F(3x3,2x2) weight gradient
UPS_ISSUED = 4.5 / 27.0            # collapsed taps behind a nearest x2 upsample (all three passes)


def step_flops_per_volume():
    """(algorithmic, issued) matrix-core FLOPs of one E+D iteration per volume for the z=1200 model: the
    wide 3x3x3 convolutions of 5 encoder + 8 decoder forward passes, 5 + 7 input-gradient passes and
    3 + 4 weight-gradient passes (utils/my_trainer.py:242-325); the single-channel convolutions and the
    1x1x1 layers (2 % of the FLOPs, HBM bound) are left out of both numbers."""
    v0 = DIMS[0] * DIMS[1] * DIMS[2]
    v1, v2, v3 = v0 // 8, v0 // 64, v0 // 512
    enc = [(64, 64, v0, 1), (64, 64, v1, 1), (64, 128, v1, 1), (128, 128, v2, 1), (128, 256, v2, 1),
           (256, 256, v3, 1), (256, 256, v3, 1), (256, 256, v3, 1)]
    dec = [(256, 256, v3, 1), (256, 256, v3, 1), (256, 256, v3, 1), (256, 128, v2, 2), (128, 128, v2, 1),
           (128, 64, v1, 2), (64, 64, v1, 1), (64, 64, v0, 2)]
    alg = iss = 0.0
    for layers, nf, nd, nw in ((enc, 5, 5, 3), (dec, 8, 7, 4)):
        for cin, cout, vox, ups in layers:
            f = 2.0 * 27 * cin * cout * vox
            alg += f * (nf + nd + nw)
            # non-upsampled input gradients at the 80x96x80 and 40x48x40 levels run the F(2,3) x F(4,3) form of
            # the Winograd kernel by default: 9 instead of 12 of the 27 multiply-adds
            dg = 9.0 / 27.0 if (ups == 1 and vox >= v1 and os.environ.get("SIV_W2_W4", "2") in ("1", "2")) else WINO2D_ISSUED
            iss += f * (nf * (UPS_ISSUED if ups == 2 else WINO2D_ISSUED) + nd * (UPS_ISSUED if ups == 2 else dg)
                        + nw * (UPS_ISSUED if ups == 2 else WGRAD_ISSUED))
    return alg, iss


def build_model(device, model_cfg=MODEL):
    import siv.models.models as models
    import siv.utils.my_trainer as trainer
    torch.manual_seed(82)                                  # z-1200main.py:152
    net = models.SoftIntroVAE(*model_cfg)
    torch.manual_seed(77)                                  # utils/my_trainer.py:160-171
    net.apply(trainer.init_weights_he)
    return net.to(device).train()


def cpu_baseline():
    """The CPU oracle (a port: plain torch fp32 on the host cores) running the SAME iteration -- one full
    E-step + D-step with both Adam updates of the z=1200 model at 80x96x80 -- at batch 1: one warm-up
    iteration (oneDNN primitive creation), then the median of three.  Threads: the job's CPU share
    (OMP_NUM_THREADS when the launcher sets it, else every core)."""
    from oracle import siv_oracle as O
    import siv.models.models as models
    import siv.utils.my_trainer as trainer
    host_cores = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or host_cores
    threads = min(threads, host_cores)
    torch.set_num_threads(threads)
    torch.manual_seed(82)
    net = models.SoftIntroVAE(*MODEL)
    torch.manual_seed(77)
    net.apply(trainer.init_weights_he)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    cfg, hp = O.Cfg(*MODEL), O.Hyper(dims=DIMS)
    gen = torch.Generator().manual_seed(103)
    ae, ad = {}, {}
    times = []
    for it in range(4):
        x = torch.rand(1, 1, *DIMS, generator=gen)
        noise = torch.randn(1, 1, DIMS[0] // 8, DIMS[1] // 8, DIMS[2] // 8, generator=gen)
        t0 = time.time()
        r = O.soft_intro_step(sd, cfg, x, noise, O.Draws(seed=it), hp, ae, ad)
        times.append(time.time() - t0)
        assert bool(torch.isfinite(r["lossE"])) and bool(torch.isfinite(r["lossD"]))
    t = sorted(times[1:])[1]
    return {"value": 1.0 / t, "unit": "volumes/s", "cores": threads, "host_cpu_count": host_cores, "kind": "port",
            "s_per_step": round(t, 3),
            "sample": f"the full E-step + D-step (+ 2 Adam updates) of the z=1200 model on ONE 80x96x80 volume "
                      f"(batch 1), oracle/siv_oracle.py on {threads} threads of {host_cores} host CPUs: 1 warm-up "
                      f"({times[0]:.1f} s), then median of 3 = {t:.2f} s per step"}


def fc_config_entry(device, steps=5, warmup=2):
    """BASELINE configs[1] beside the headline number (not part of it): the FC-latent z=600 model
    mymodel.SoftIntroVAE(16,32,64,128,600), batch 4, one GPU -- iteration time, and its Linear(19200 -> 1200)
    forward against the HBM peak (the layer streams its 92 MB weight matrix once; with 4 batch rows on a
    16-row matrix-core tile the matrix pipe is never the limit: its FLOP rate is reported, not a roofline)."""
    import siv.backend as sb
    import siv.models.mymodel as mymodel
    import siv.rng as srng
    import siv.utils.my_trainer as trainer
    from siv.optim import FlatAdam
    batch = 4
    torch.manual_seed(82)
    net = mymodel.SoftIntroVAE(16, 32, 64, 128, 600)
    torch.manual_seed(77)
    net.apply(trainer.init_weights_he)
    net = net.to(device).train()
    opt_e = FlatAdam(net.encoder.parameters(), lr=2e-4)
    opt_d = FlatAdam(net.decoder.parameters(), lr=2e-4)
    step = trainer.SoftIntroStep(net, opt_e, opt_d, 1.0, 1024.0, 0.75, 1e-8, None, None)
    gen = torch.Generator().manual_seed(103)
    real = [torch.rand(batch, 1, *DIMS, generator=gen).to(device) for _ in range(2)]
    for i in range(warmup + steps):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        st = step(real[i % 2], srng.get_rng().eps((batch, net.z_ch), device))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert bool(torch.isfinite(st).all())
    B = sb.get_backend()
    K, O = 19200, 1200
    x = torch.randn(batch, K, device=device)
    w = torch.randn(O, K, device=device) * K ** -0.5
    B.linear_fwd(x, w, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        B.linear_fwd(x, w, None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    byts = (O * K + batch * K + batch * O) * 4
    return {"workload": "BASELINE configs[1]: 80x96x80 fp32, z=600, mymodel.SoftIntroVAE(16,32,64,128,600), batch 4",
            "ms_per_step": round(dt * 1e3, 2), "volumes_per_s": round(batch / dt, 2), "steps": steps,
            "linear_19200_to_1200_fwd": {"bound": "hbm", "achieved": round(byts / (ms * 1e-3) / 1e9, 1),
                                         "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s",
                                         "frac": round(byts / (ms * 1e-3) / 1e9 / (HBM_PEAK_TBS * 1e3), 4),
                                         "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": byts,
                                         "mfma_tflops_issued": round(2.0 * 16 * K * O / (ms * 1e-3) / 1e12, 2),
                                         "mfma_frac_of_fp32_peak": round(2.0 * 16 * K * O / (ms * 1e-3) / 1e12
                                                                         / FP32_PEAK_TFLOPS, 4)}}


def main():
    args = parse_args()
    rehearse = args.rehearse_cpu
    if args.batch is None:
        args.batch = 2 if rehearse else PER_GPU_BATCH
    model_cfg, dims = (REHEARSAL_MODEL, REHEARSAL_DIMS) if rehearse else (MODEL, DIMS)

    # keep stdout for the one JSON line: library banners (RCCL prints its version on the first
    # communicator) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import siv.backend as sb
    import siv.rng as srng
    import siv.utils.my_trainer as trainer
    from siv.optim import FlatAdam
    from siv.parallel import GradSync, init_process_group

    if rehearse:
        # launcher / rank-code rehearsal on CPU: the arithmetic comes from an injected backend class
        spec = os.environ.get("SIV_BENCH_BACKEND")
        if not spec:
            raise SystemExit("--rehearse-cpu needs SIV_BENCH_BACKEND=module:Class (there is no built-in CPU path)")
        import importlib
        mod, cls = spec.split(":")
        sb.set_backend(getattr(importlib.import_module(mod), cls)())
        torch.set_num_threads(2)
    rank, local_rank, world = init_process_group("gloo" if rehearse else "nccl")
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if rehearse:
        device = torch.device("cpu")
    else:
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
    B = sb.get_backend()
    assert rehearse or B.name == "hip"

    net = build_model(device, model_cfg)
    sync = GradSync(device)
    sync.broadcast_(net)                                   # replicas start from rank 0's parameters and buffers
    opt_e = FlatAdam(net.encoder.parameters(), lr=2e-4)
    opt_d = FlatAdam(net.decoder.parameters(), lr=2e-4)
    step = trainer.SoftIntroStep(net, opt_e, opt_d, 1.0, 1024.0, 0.75, 1e-8, None, sync)
    if rehearse:
        class _HostDraws:                                  # explicit host tensors in place of the device generator
            g = torch.Generator().manual_seed(77 + rank)

            def mask(self, shape, p):
                return (torch.rand(shape, generator=self.g) >= p).float()

            def eps(self, shape):
                return torch.randn(shape, generator=self.g)
        srng.set_rng(srng.ReplayRng(_HostDraws()))
    else:
        srng.manual_seed(77, rank)

    gen = torch.Generator().manual_seed(103 + rank)        # data seed, z-1200main.py:63
    n_batches = 2
    batches = [torch.rand(args.batch, 1, *dims, generator=gen).to(device) for _ in range(n_batches)]

    # live timing of the dominant kernel: HIP events on the launch stream around every forward
    # launch of the 64->64 conv at full resolution inside the timed region
    dom_events = []
    recording = {"on": False}

    def timed(orig):
        def call(x, w, bias, ups=1, *rest):
            hit = (recording["on"] and ups == 1 and w.shape[0] == 64 and w.shape[1] == 64
                   and tuple(x.shape[2:]) == DIMS)
            if not hit:
                return orig(x, w, bias, ups, *rest)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(x, w, bias, ups, *rest)
            e1.record()
            dom_events.append((e0, e1, x.shape[0]))
            return out
        return call

    # the layer runs through conv3d_k3_fwd_stats in training mode (convolution + BatchNorm statistics
    # in one C call: the event pair then also spans the weight-packing and the statistics-finalize
    # launches, ~15 us of 4.5 ms) and through conv3d_k3_fwd otherwise
    if not rehearse:
        B.conv3d_k3_fwd = timed(B.conv3d_k3_fwd)
        B.conv3d_k3_fwd_stats = timed(B.conv3d_k3_fwd_stats)

    # ... and three HBM-bound members of the step: the BatchNorm+LeakyReLU(+AvgPool) backward of a 64-channel
    # full-resolution layer, the stem's fused BatchNorm backward and the decoder's 64 -> 1 output convolution with
    # its BatchNorm prologue.  They are timed in two extra iterations AFTER the timed region with the
    # weight gradients kept on the main stream, so that nothing shares the chip with them (inside the
    # timed region a side-stream weight gradient time-shares the CUs and the event pair would measure that).
    # Each entry carries the algorithmic bytes (every tensor once) and the bytes its passes really move.
    hbm_events = {"bn_act_bwd": [], "stem_bn_bwd": [], "stem_wgrad": [], "conv_64_to_1": []}
    hbm_on = {"on": False}

    def timed_hbm(orig, name, hit_fn, bytes_fn):
        def call(*a, **k):
            if not (hbm_on["on"] and hit_fn(*a)):
                return orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **k)
            e1.record()
            hbm_events[name].append((e0, e1) + tuple(bytes_fn(*a)))
            return out
        return call

    if not rehearse:
        B.bn_act_bwd = timed_hbm(
            B.bn_act_bwd, "bn_act_bwd",
            # args after (dout, y): mean, var, gamma, beta, res, keep, p, slope, pool, eps, ...
            # the pooled BatchNorm behind the first wide encoder convolution: dout at 40x48x40, y and dy at
            # 80x96x80.  Algorithmic: y and dout read, dy written.  Moved: the reduction pass reads y and dout,
            # the apply pass reads them again and writes dy.
            lambda dout, y, *r: y.shape[1] == 64 and tuple(y.shape[2:]) == DIMS and r[8] == 2 and r[4] is None,
            lambda dout, y, *r: (2 * y.numel() * 4 + dout.numel() * 4, 3 * y.numel() * 4 + 2 * dout.numel() * 4))
        # the encoder stem's BatchNorm backward (its convolution output is recomputed, not read).  Algorithmic:
        # dout (and a stored mask) read, the gradient of the convolution output written.  Moved: dout (and the
        # mask) read by both passes.
        B.conv1_bnact_bwd = timed_hbm(
            B.conv1_bnact_bwd, "stem_bn_bwd",
            lambda dout, *r: dout.shape[1] == 64 and tuple(dout.shape[2:]) == DIMS,
            lambda dout, x, w, bias, mean, var, gamma, beta, keep, *r: (
                2 * dout.numel() * 4 + (dout.numel() if torch.is_tensor(keep) else 0),
                3 * dout.numel() * 4 + (2 * dout.numel() if torch.is_tensor(keep) else 0)))
        # the stem's backward where its input needs no gradient (the E-step's three encoder passes): weight,
        # bias and BatchNorm-affine gradients from one pass over dout plus the decision bits of the forward pass
        # (1/16 of dout's bytes) and two passes over the single-channel input; nothing written
        B.conv1_bnact_wgrad = timed_hbm(
            B.conv1_bnact_wgrad, "stem_wgrad",
            lambda dout, *r: dout.shape[1] == 64 and tuple(dout.shape[2:]) == DIMS,
            lambda dout, x, *r: (dout.numel() * 4 + dout.numel() // 4 + 2 * x.numel() * 4,) * 2)
        # (the output convolution applies the BatchNorm + LeakyReLU of the block in front while it loads:
        # the raw 64-channel tensor read once, one channel written; one pass)
        B.bnact_conv_to1_fwd = timed_hbm(
            B.bnact_conv_to1_fwd, "conv_64_to_1",
            lambda x, *r: x.shape[1] == 64 and tuple(x.shape[2:]) == DIMS,
            lambda x, *r: ((x.numel() + x.numel() // 64) * 4, (x.numel() + x.numel() // 64) * 4))

    graphed = None

    def one_step(i):
        real = batches[i % n_batches]
        noise = srng.get_rng().eps(net.latent_shape(real.size(0), dims), device)
        return (graphed or step)(real, noise)

    multi = dist.is_initialized()       # world > 1, or a forced one-rank rehearsal of the RCCL path

    def barrier():
        if multi:
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    barrier()
    if args.graph and not multi:
        # kernel-level timing needs individually launched kernels: two eager iterations first
        recording["on"] = True
        for i in range(2):
            one_step(i)
        barrier()
        recording["on"] = False
        graphed = trainer.GraphedSoftIntroStep(step, batches[0].shape, net.latent_shape(args.batch, dims), device)
        one_step(0)
        barrier()
    recording["on"] = graphed is None
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = one_step(args.warmup + i)
    host_elapsed = time.perf_counter() - t0     # Python + launch time only: nothing above waits for the GPU
    barrier()
    elapsed = time.perf_counter() - t0
    recording["on"] = False
    if multi:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = dict(zip(trainer.STAT_NAMES, last.cpu().tolist()))
    if not all(v == v for v in stats.values()):
        raise SystemError(f"NaN in losses: {stats}")

    if not rehearse:
        # the HBM-bound members, alone on the chip (see above): two more iterations, not part of `value`
        import siv.ops as sops
        saved_overlap = sops.OVERLAP_WGRAD
        sops.OVERLAP_WGRAD = False
        hbm_on["on"] = True
        for i in range(2):
            step(batches[i % n_batches], srng.get_rng().eps(net.latent_shape(args.batch, dims), device))
        barrier()
        hbm_on["on"] = False
        sops.OVERLAP_WGRAD = saved_overlap

    # Second, separately reported measurement: the same iteration with the opt-in kernels that form
    # their fp32 products from bf16 triples on the bf16 matrix cores (conv path 5).  It is NOT the
    # headline value: the headline runs the exact-fp32 matrix-core kernels.
    split = None
    if args.split_bf16:
        B.set_conv_path(5)
        k2 = max(2, min(args.steps, 5))
        one_step(0)
        barrier()
        t1 = time.perf_counter()
        for i in range(k2):
            last2 = one_step(1 + i)
        barrier()
        el2 = time.perf_counter() - t1
        B.set_conv_path(0)
        if multi:
            t = torch.tensor([el2], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        st2 = dict(zip(trainer.STAT_NAMES, last2.cpu().tolist()))
        split = {"value": round(args.batch * world * k2 / el2, 4), "unit": "volumes/s", "steps": k2,
                 "ms_per_step": round(1000.0 * el2 / k2, 2), "finite": all(v == v for v in st2.values()),
                 "what": "same iteration with siv_set_conv_path(5): Winograd and upsampled-conv forward / input-"
                         "gradient kernels form each fp32 product from three bf16 pieces per operand on "
                         "v_mfma_f32_32x32x16_bf16 (six piece products, fp32 accumulate; per-layer error vs fp64 "
                         "at or below that of the fp32 kernels); weight gradients and everything else unchanged"}

    # Third, separately reported and NOT the headline either: the same iteration with the FORWARD convolutions of
    # the 80x96x80 and 40x48x40 levels on the F(2,3) x F(4,3) form of the Winograd kernel too (SIV_W2_W4=1; the
    # default applies it to the input gradients only).  All fp32, per-layer rounding error that of the direct
    # kernel, but at batch 1 the end-to-end gradient bounds of tests/test_gpu_fullsize.py are exceeded with it.
    f2x4 = None
    if world == 1 and not rehearse and not args.no_f2x4 and graphed is None:
        w4_before = os.environ.get("SIV_W2_W4")
        os.environ["SIV_W2_W4"] = "1"
        try:
            k3 = max(2, min(args.steps, 5))
            one_step(0)
            barrier()
            t1 = time.perf_counter()
            for i in range(k3):
                last3 = one_step(1 + i)
            barrier()
            el3 = time.perf_counter() - t1
        finally:
            if w4_before is None:
                del os.environ["SIV_W2_W4"]
            else:
                os.environ["SIV_W2_W4"] = w4_before
        st3 = dict(zip(trainer.STAT_NAMES, last3.cpu().tolist()))
        f2x4 = {"value": round(args.batch * k3 / el3, 4), "unit": "volumes/s", "steps": k3,
                "ms_per_step": round(1000.0 * el3 / k3, 2), "finite": all(v == v for v in st3.values()),
                "what": "same iteration with SIV_W2_W4=1: the forward convolutions of the 80x96x80 and 40x48x40 levels "
                        "on Winograd F(2,3) x F(4,3) too (the default applies it to the input gradients only); fp32 "
                        "MFMA, 9 multiply-adds per output instead of 12; opt-in, not the headline"}

    if rank == 0:
        ms_per_step = 1000.0 * elapsed / args.steps
        vols = args.batch * world * args.steps / elapsed
        dom_ms = [e0.elapsed_time(e1) for e0, e1, _ in dom_events]
        roof = None
        alg_step, iss_step = step_flops_per_volume()
        if dom_ms:
            avg_ms = sum(dom_ms) / len(dom_ms)
            algorithmic = DOM_FLOP_PER_VOLUME * args.batch / (avg_ms * 1e-3) / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            # The kernel applies Winograd F(2x2,3x3) over (H, W): it issues 12/27 of the direct-convolution
            # multiply-adds.  `achieved` / `frac` are the matrix-core FLOPs actually ISSUED against the fp32
            # MFMA peak (<= 1); `algorithmic_tflops` is the direct-convolution-equivalent rate.
            achieved = algorithmic * WINO2D_ISSUED
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_PEAK_TFLOPS, 4), "traffic": traffic,
                    "algorithmic_tflops": round(algorithmic, 2),
                    "algorithmic_bytes_per_launch": int((2 * 64 * 614400 * args.batch + 27 * 64 * 64) * 4),
                    "kernel": "k3_fwd_wino2d (3x3x3 conv 64->64 @80x96x80, Winograd F(2x2,3x3) over (H,W) on fp32 MFMA "
                              "32x32x2: 12 instead of 27 multiply-adds per output and channel pair)",
                    "launches_timed": len(dom_ms), "avg_launch_ms": round(avg_ms, 3)}
        step_s = elapsed / args.steps
        roof_step = {"bound": "mfma", "unit": "TFLOP/s", "peak": FP32_PEAK_TFLOPS,
                     "achieved": round(iss_step * args.batch / step_s / 1e12, 2),
                     "frac": round(iss_step * args.batch / step_s / 1e12 / FP32_PEAK_TFLOPS, 4),
                     "algorithmic_tflops": round(alg_step * args.batch / step_s / 1e12, 2),
                     "what": "whole iteration: matrix-core FLOPs issued by the wide 3x3x3 convolutions of all 13 forward, "
                             "12 input-gradient and 7 weight-gradient passes / step time (everything else in the step "
                             "counts as time only)"}
        roof_hbm = {}
        for name, evs in hbm_events.items():
            if evs:
                ms = [e[0].elapsed_time(e[1]) for e in evs]
                avg = sum(ms) / len(ms)
                byts = sum(e[2] for e in evs) / len(evs)
                moved = sum(e[3] for e in evs) / len(evs)
                gbs = byts / (avg * 1e-3) / 1e9
                roof_hbm[name] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s",
                                  "frac": round(gbs / (HBM_PEAK_TBS * 1e3), 4), "launches_timed": len(ms),
                                  "avg_launch_ms": round(avg, 3),
                                  "algorithmic_bytes_per_launch": int(byts),
                                  "moved_bytes_per_launch": int(moved),
                                  "moved_gbs": round(moved / (avg * 1e-3) / 1e9, 1),
                                  "timed": "two iterations after the timed region, weight gradients on the main "
                                           "stream (nothing else on the chip)"}
        out = {
            "metric": "MRI volumes/sec (S-IntroVAE step) at 80x96x80 z=1200",
            "value": round(vols, 4), "unit": "volumes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 80x96x80 fp32, z=1200, "
                                   "SoftIntroVAE(64,[[64,1,2],[128,1,2],[256,2,2]]), E-step + D-step + 2 Adam updates",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       "rng": "device Philox (dropout, eps, noise)",
                       "launch": "HIP graph replay" if graphed is not None else "kernel by kernel"},
            "host_enqueue_ms_per_step": round(1000.0 * host_elapsed / args.steps, 2),
            "roofline": roof,
            "roofline_step": roof_step,
            "roofline_hbm": roof_hbm,
            "split_bf16_products": split,
            "winograd_f2x4": f2x4,
            "losses": {k: stats[k] for k in ("lossE", "lossD", "loss_rec", "kl_real")},
        }
        if rehearse:
            # a rehearsal of the launcher and the rank code, not a measurement of anything
            out["metric"] = "REHEARSAL (CPU, gloo, narrow model, 16x24x16): not a measurement"
            out["data"] = "synthetic (rehearsal)"
            out["config"]["workload"] = "launcher rehearsal: SoftIntroVAE(4,[[4,1,2],[8,1,2],[8,2,2]]) at 16x24x16"
            out["config"]["rng"] = "host tensors"
            for k in ("roofline", "roofline_step", "roofline_hbm"):
                out[k] = None
        if world == 1 and not args.no_fc and not rehearse:
            out["fc_config"] = fc_config_entry(device)
        if world == 1 and not args.no_cpu_baseline and not rehearse:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
