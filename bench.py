#!/usr/bin/env python3
"""bench.py - training samples/sec of the MegaCRN hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = the reference trainer's optimizer step (model/traintest_MegaCRN.py:115-130): forward,
3-term loss, backward, one all-reduce of the flat gradient bucket (N>1), clip_grad_norm_(5), Adam.
Inputs are synthetic and already resident in HBM when the timed region starts.
Weak scaling: every rank processes a full config batch.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "metrla":  dict(N=207, B=64, T=12, H=64, M=20, D=64, label="METR-LA-shaped"),
    "pemsbay": dict(N=325, B=64, T=12, H=64, M=20, D=64, label="PEMS-BAY-shaped"),
    "expytky": dict(N=1843, B=32, T=6, H=32, M=10, D=32, label="EXPY-TKY-shaped"),
    "syn8192": dict(N=8192, B=32, T=12, H=64, M=20, D=64, label="synthetic N=8192"),
}
SC_MEAN, SC_STD = 54.4, 19.5
PEAK = {"f32": 157.3e12, "bf16x3": 2500e12, "bf16": 2500e12}
HBM_PEAK = 8.0e12                              # MI355X_MICROARCH.md: HBM3E 8 TB/s   # MI355X_MICROARCH.md dense MFMA peaks: fp32 / bf16
MAX_CLOCK_MHZ = 2400.0                          # MI355X_MICROARCH.md: max clock, the clock the dense peaks are quoted at
DEFAULT_PREC = {"metrla": "bf16x3", "pemsbay": "bf16x3", "expytky": "bf16", "syn8192": "bf16"}
ROLE_NAMES = ["misc", "propagate", "weight_pool", "dgrad", "propagate_T", "adjacency_grad", "weight_grad", "propagate_inputs"]


def synth(cfg, B, seed, device):
    """x, labels ~ N(0,1) with 8% standardized zeros; ycov = time-of-day ramp."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    N, T = cfg["N"], cfg["T"]
    miss = (0.0 - SC_MEAN) / SC_STD
    x = torch.randn(B, T, N, 1, generator=g)
    y = torch.randn(B, T, N, 1, generator=g)
    x[torch.rand(x.shape, generator=g) < 0.08] = miss
    y[torch.rand(y.shape, generator=g) < 0.08] = miss
    t0 = torch.randint(0, 288, (B, 1, 1, 1), generator=g).float()
    ycov = ((t0 + T + torch.arange(T).view(1, T, 1, 1)) / 288.0) % 1.0
    ycov = ycov.expand(B, T, N, 1).contiguous()
    return x.to(device), ycov.to(device), y.to(device)


def alg_flops_forward(cfg, B):
    """Algorithmic FLOPs of one forward (BASELINE.md section 4 / SURVEY.md 8(d)); train step = 3x."""
    N, T, H, M, D, K = cfg["N"], cfg["T"], cfg["H"], cfg["M"], cfg["D"], 3
    Hd = H + D

    def agcn(Cc, O):
        return 2 * (K - 1) * 2 * N * N * B * Cc + 2 * B * N * (2 * K * Cc) * O

    f = T * (agcn(1 + H, 2 * H) + agcn(1 + H, H)) + T * (agcn(2 + Hd, 2 * Hd) + agcn(2 + Hd, Hd))
    f += 2 * (2 * N * M * D) + 2 * (2 * N * N * D) + 2 * B * N * H * D + 4 * B * N * D * M + T * 2 * B * N * Hd
    return float(f)


PROFILES_DIR = os.path.join(ROOT, "profiles")


def lib_identity():
    """what a measured artefact must name to be believed: the build id of the library this process loaded"""
    from megacrn_amd import _lib
    return _lib.build_id()


def traffic_profile(config_name, dtype, profiles_dir=None, build_id=None):
    """(table, source) - the committed PMC traffic table of this config and arithmetic (tools/pmc_traffic.sh): per kernel the
    FETCH_SIZE x 2 + WRITE_SIZE bytes per launch averaged over the dispatches of ONE steady-state train step (the window
    between the last two k_clip_adam dispatches of the run: no autotuner candidates, no warm-up), and `_step` = their sum.
    The table of the NEWEST profiles/r<NN>/ that holds one is taken, and only if it was measured on THIS build of the library
    (`_meta.build_id` == mcrn_build_id()): bytes of another build divided by this run's time would be a number about nothing.
    table is None when no table exists or the newest one is stale; source says which ({"file", "status", ...})."""
    profiles_dir = profiles_dir or PROFILES_DIR
    build_id = build_id if build_id is not None else lib_identity()
    suffix = "" if dtype == DEFAULT_PREC[config_name] else f"_{dtype}"
    rounds = []
    for d in os.listdir(profiles_dir) if os.path.isdir(profiles_dir) else []:
        if d[:1] == "r" and d[1:].isdigit() and os.path.exists(os.path.join(profiles_dir, d, f"traffic_{config_name}{suffix}.json")):
            rounds.append((int(d[1:]), d))
    if not rounds:
        return None, {"status": "absent"}
    rd = max(rounds)[1]
    path = os.path.join(profiles_dir, rd, f"traffic_{config_name}{suffix}.json")
    d = json.load(open(path))
    meta = d.get("_meta", {})
    src = {"file": os.path.relpath(path, os.path.dirname(profiles_dir)), "measured_on_build": meta.get("build_id"),
           "this_build": build_id, "git": meta.get("git")}
    if "_step" not in d or meta.get("build_id") != build_id:
        src["status"] = "stale: measured on another build of the library, not reported"
        return None, src
    src["status"] = "current"
    return d, src


def x3r_session(cfg, B, dtype):
    """engine.hip plan_model, ModelPlan::x3r: a bf16x3 session on a graph beyond the fused two-hop kernels (N > 352) runs the bf16 mode's
    data flow with hi/lo operand pairs - three MFMAs per product on the bf16-resident GEMM - when the hoisted forward and backward take
    the shape (H % 32 == 0, B x input channels % 8 == 0)."""
    return dtype == "bf16x3" and cfg["N"] > 352 and cfg["H"] % 32 == 0 and (cfg["H"] + cfg["D"]) % 32 == 0 and B % 8 == 0


def pmc_traffic(config_name, dtype, N, x3r=False):
    """Fabric-side bytes per launch of the dominant kernel (the forward propagation) in a steady-state step:
    FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes, FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md."""
    d, _ = traffic_profile(config_name, dtype)
    if d is None:
        return None
    # (tools/pmc_traffic.sh folds the tile parameters of the tuned GEMMs away: family<*, operand forms, ROLE>)
    if dtype == "bf16" or x3r:
        keys = ("gemm_bf16*<*, true, 1>",)        # gemm_bf16(_pp)_kernel<..., BTR = true, ROLE = 1>
    elif N <= 352 and dtype == "bf16x3":
        keys = ("prop2_fwd_kernel",)              # fused two-hop kernels (prop_small.h)
    else:
        keys = ("<*, true, false, 1>",)           # tiled gemm_*_kernel<..., AKC, !BKC, ROLE = 1>
    tot = n = 0
    for k, v in d.items():
        if k[:1] != "_" and any(key in k for key in keys):
            tot += v["hbm_bytes_per_launch_corrected"] * v["launches"]
            n += v["launches"]
    return round(tot / n) if n else None


def step_traffic(config_name, dtype, ms_per_step):
    """Bytes one steady-state train step moves through the L2s' fabric ports (the sum of the committed PMC table over the step's
    launches; Infinity-Cache hits are counted, MI355X_MICROARCH.md) and the rate that is at this run's step time."""
    d, src = traffic_profile(config_name, dtype)
    if d is None:
        return {"traffic_source": src}
    gb = d["_step"]["fabric_bytes"] / 1e9
    return {"step_fabric_gb": round(gb, 3), "step_fabric_tbs": round(gb / ms_per_step, 3), "traffic_source": src,
            "step_fabric_note": f"sum over the {d['_step']['dispatches']} dispatches of one steady-state step of (2 x FETCH_SIZE + WRITE_SIZE), "
                                f"collected by tools/pmc_traffic.sh in its own rocprofv3 --pmc runs on this build of the library "
                                f"({src['file']}); / this run's ms_per_step"}


def small_hoist_fwd(cfg, B):
    """engine.hip plan_model, Shp::hoist_fwd: the decoder's per-step propagation covers the B*H_dec state columns only where that
    saves a pass of the fused two-hop kernels (PEMS-BAY at B = 64; not METR-LA, whose full width is one pass of 96-column units)."""
    env = os.environ.get("MCRN_HOIST_FWD", "1")
    N, Hd = cfg["N"], cfg["H"] + cfg["D"]
    if env == "0" or N > 352 or Hd % 64:
        return False
    cdiv = lambda a, b: -(-a // b)
    ld = B * ((Hd + 2 + 3) // 4 * 4)
    full = 1 if ((N + 31) // 32 <= 8 and cdiv(ld, 96) <= 128) else cdiv(cdiv(ld, 64), 128)
    return env == "2" or cdiv(B * (Hd // 64), 128) < full


def propagation_alg_bytes(cfg, B, dtype):
    """Algorithmic HBM bytes of ONE forward propagation launch, averaged over the encoder and decoder launches of a
    step like the flops (read the supports once, the input plane once, write the propagated planes)."""
    N, H, D, K = cfg["N"], cfg["H"], cfg["D"], 3
    out = []
    for C, Hs in ((1 + H, H), (2 + H + D, H + D)):
        if x3r_session(cfg, B, dtype):
            # hi/lo operand pairs: stacked adjacency and the packed state block twice (hi + lo images), fp32 planes out
            out.append(2 * (2 * (K - 1) * N * N * 2 + N * B * Hs * 2) + 2 * (K - 1) * N * B * Hs * 4)
        elif dtype == "bf16" and Hs in (32, 64, 128):
            # hoisted bf16 mode: stacked bf16 adjacency (2(K-1) blocks), the bf16 state block (Hs channels;
            # the input channels are propagated once per stack), bf16-resident output planes
            out.append(2 * (K - 1) * N * N * 2 + N * B * Hs * 2 + 2 * (K - 1) * N * B * Hs * 2)
        elif dtype == "bf16":     # stacked bf16 adjacency, bf16 input plane, fp32 output planes
            out.append(2 * (K - 1) * N * N * 2 + N * B * C * 2 + 2 * (K - 1) * N * B * C * 4)
        elif Hs == H + D and small_hoist_fwd(cfg, B):
            # decoder of the small graphs, forward hoisting (engine.hip Shp::hoist_fwd): the per-step launch covers the state channels
            out.append(2 * N * N * 4 + N * B * Hs * 4 + 2 * (K - 1) * N * B * Hs * 4)
        else:                     # fp32 storage: both supports, input plane, 2(K-1) output planes
            out.append(2 * N * N * 4 + N * B * C * 4 + 2 * (K - 1) * N * B * C * 4)
    per_call = sum(out) / 2.0
    # the tiled path at N > 352 (f32; bf16x3 shapes the resident path does not take) launches one hop at a time: half the bytes per launch
    hops_per_launch = 1 if (dtype != "bf16" and N > 352 and not x3r_session(cfg, B, dtype)) else 2
    return per_call * hops_per_launch / 2.0


def cpu_model_string():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_numpy(cfg, sample_B, sample_T, nthr):
    """The numpy oracle (the parity checker) timed on a bounded sample: one warm step on 2 samples, then one timed train
    step of `sample_B` samples x `sample_T` (<= T) sequence steps on `nthr` BLAS threads; samples/s scaled to T."""
    from oracle import megacrn_oracle as O
    from threadpoolctl import threadpool_limits
    N, T, H = cfg["N"], cfg["T"], cfg["H"]
    P = O.init_params(N, rnn_units=H, mem_num=cfg["M"], mem_dim=cfg["D"], seed=0)
    rng = np.random.default_rng(0)
    mk = lambda b: (rng.standard_normal((b, sample_T, N, 1)).astype(np.float32), rng.random((b, sample_T, N, 1)).astype(np.float32),
                    rng.standard_normal((b, sample_T, N, 1)).astype(np.float32))
    opt = O.Adam(P)
    teacher = [i % 2 == 0 for i in range(sample_T)]
    with threadpool_limits(limits=nthr):
        O.train_step(P, opt, *mk(min(2, sample_B)), teacher, SC_MEAN, SC_STD)              # warm: BLAS pool, page-in
        x, yc, y = mk(sample_B)
        t = time.perf_counter()
        O.train_step(P, opt, x, yc, y, teacher, SC_MEAN, SC_STD)
        dt = time.perf_counter() - t
    return sample_B / dt * (sample_T / T), dt


def cpu_baseline_torch(cfg, sample_B, sample_T, nthr):
    """The reference's own ATen op sequence on the host cores (oracle/megacrn_torch_cpu.py: einsum / cat / sigmoid ...,
    autograd backward, clip_grad_norm_, torch Adam - SURVEY.md 8(d)), one warm step on 2 samples then one timed train
    step of `sample_B` samples x `sample_T` sequence steps at `nthr` threads; samples/s scaled linearly to T."""
    from oracle import megacrn_oracle as O
    from oracle import megacrn_torch_cpu as TC
    N, T, H = cfg["N"], cfg["T"], cfg["H"]
    P = TC.make_params(O.init_params(N, rnn_units=H, mem_num=cfg["M"], mem_dim=cfg["D"], seed=0))
    opt = torch.optim.Adam(list(P.values()), lr=0.01, eps=1e-3)
    g = torch.Generator().manual_seed(0)
    mk = lambda b: (torch.randn(b, sample_T, N, 1, generator=g), torch.rand(b, sample_T, N, 1, generator=g),
                    torch.randn(b, sample_T, N, 1, generator=g))
    teacher = [i % 2 == 0 for i in range(sample_T)]
    prev = torch.get_num_threads()
    torch.set_num_threads(nthr)
    try:
        TC.train_step(P, opt, *mk(min(2, sample_B)), teacher, SC_MEAN, SC_STD)             # warm
        x, yc, y = mk(sample_B)
        t = time.perf_counter()
        TC.train_step(P, opt, x, yc, y, teacher, SC_MEAN, SC_STD)
        dt = time.perf_counter() - t
    finally:
        torch.set_num_threads(prev)
    return sample_B / dt * (sample_T / T), dt


def cpu_sample(cfg, B, budget_flops):
    """(samples, sequence length) of a CPU leg so that one train step stays within `budget_flops` algorithmic flops:
    the full batch when it fits, else as many whole samples as fit, else one sample with a shortened sequence."""
    per_sample = 3.0 * alg_flops_forward(cfg, 1)
    nb = int(budget_flops // per_sample)
    if nb >= 1:
        return min(nb, B), cfg["T"]
    return 1, max(1, min(cfg["T"], int(budget_flops // (per_sample / cfg["T"]))))


def launch_cmd(gpus, port, argv):
    """torchrun command line of the N ranks (one per GPU, rendezvous on 127.0.0.1: the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_env():
    """Environment of the ranks.  HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC;
    with the legacy mode RCCL's cross-process buffer registration fails (hipIpcGetMemHandle: invalid argument).  The
    images export it already; set here only if the caller's environment lacks it."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves, as children,
    BEFORE this process touches the GPU (a process that initialised HIP must never re-exec), and pass their exit
    code on.  One rank per GPU, rendezvous on 127.0.0.1, backend nccl (= RCCL over xGMI)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return subprocess.call(launch_cmd(args.gpus, port, sys.argv[1:]), env=launch_env())


def prop_kernel_name(cfg, dtype, B=8):
    small = cfg["N"] <= 352 and dtype == "bf16x3"
    return ("mcrn::gemm_bf16(_pp)_kernel<BM,BN,..,BTR=true,ROLE=1> (all Chebyshev terms of both supports, one product)" if dtype == "bf16" else
            "mcrn::gemm_bf16(_pp)_kernel<BM,BN,..,BTR=true,ROLE=1> with hi/lo operand pairs (nterm = 3: all Chebyshev terms of both supports, "
            "one product, three MFMAs per product)" if x3r_session(cfg, B, dtype) else
            "mcrn::prop2_fwd_kernel<NF,CT> (both Chebyshev hops fused)" if small else
            "mcrn::gemm_%s_kernel<..., ROLE=1>" % ("bf16x3" if dtype == "bf16x3" else "f32"))


NO_TEACHER_STEP = 10 ** 6     # batches_seen at which cl / (cl + exp(step / cl)) is 0 (model/MegaCRN.py:146-147): no step teacher-forced


def regime_legs(tr, batch, B, steps, warmup, sync):
    """What training mostly runs (model/MegaCRN.py:146-147,188-191: the teacher-forcing probability is 0.5 at batches_seen =
    15 200 and ~0 beyond 30 000, while the headline steps start at batches_seen = 0 and are fully teacher-forced) and the
    evaluation forward of the reference's val / test passes (model/traintest_MegaCRN.py:50-99): same batch, same process."""
    x, ycov, y = batch
    saved = tr.batches_seen
    tr.batches_seen = NO_TEACHER_STEP
    for _ in range(warmup):
        tr.train_step(x, ycov, y)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(x, ycov, y)
    sync()
    dt_nt = time.perf_counter() - t0
    tr.batches_seen = saved
    m = tr.model
    m.eval()
    with torch.no_grad():
        for _ in range(warmup):
            m(x, ycov)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            m(x, ycov)
        sync()
        dt_ev = time.perf_counter() - t0
    m.train()
    return {"value_no_teacher": round(B * steps / dt_nt, 2), "ms_per_step_no_teacher": round(1e3 * dt_nt / steps, 4),
            "eval_samples_per_s": round(B * steps / dt_ev, 2), "ms_per_eval_forward": round(1e3 * dt_ev / steps, 4),
            "regime_steps": steps}


def time_role(tr, batch, role, nrep):
    """HIP events around every launch of GEMM role `role` (on the stream it is launched on) during `nrep` real train steps."""
    from megacrn_amd._lib import lib, check
    check(lib.mcrn_prof_begin(role), "prof_begin")
    for _ in range(nrep):
        tr.train_step(*batch)
    ms, n, af, ef = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
    torch.cuda.synchronize()
    check(lib.mcrn_prof_end(C.byref(ms), C.byref(n), C.byref(af), C.byref(ef)), "prof_end")
    return ms.value, n.value, af.value, ef.value


def held_clock():
    """(MHz, stamped launches): shader clock inside the K loops of the bf16-resident products of the role profiled last (in-kernel
    stamps, mcrn_prof_clock_mhz); (None, 0) for the roles that run other kernels."""
    from megacrn_amd._lib import lib
    mhz, nl = C.c_double(), C.c_longlong()
    lib.mcrn_prof_clock_mhz(C.byref(mhz), C.byref(nl))
    return (round(mhz.value, 1), nl.value) if nl.value > 0 and mhz.value > 0 else (None, 0)


def roofline_of(tr, batch, cfg, config_name, B, dtype, nrep=5):
    """`roofline` object of the forward K-hop propagation (the dominant kernel north_star names)."""
    ms, n, af, _ = time_role(tr, batch, 1, nrep)
    mhz, n_stamped = held_clock()
    launch_s = ms * 1e-3 / n
    alg_flops = af / n
    alg_bytes = propagation_alg_bytes(cfg, B, dtype)
    ai = alg_flops / alg_bytes
    ridge = PEAK[dtype] / HBM_PEAK
    frac_mfma, frac_hbm = alg_flops / launch_s / PEAK[dtype], alg_bytes / launch_s / HBM_PEAK
    bound = "mfma" if ai >= ridge else "hbm"
    return {"bound": bound, "kernel": prop_kernel_name(cfg, dtype, B) + " (K-hop propagation S x Z, model/MegaCRN.py:25)",
            "achieved": round(alg_flops / launch_s / 1e12, 3) if bound == "mfma" else round(alg_bytes / launch_s / 1e9, 1),
            "peak": round(PEAK[dtype] / 1e12, 1) if bound == "mfma" else HBM_PEAK / 1e9,
            "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
            "frac": round(frac_mfma if bound == "mfma" else frac_hbm, 5),
            "traffic": pmc_traffic(config_name, dtype, cfg["N"], x3r_session(cfg, B, dtype)),
            "traffic_source": traffic_profile(config_name, dtype)[1],
            "arithmetic_intensity": round(ai, 1), "ridge": round(ridge, 1),
            "frac_of_mfma_peak": round(frac_mfma, 5), "frac_of_hbm_peak": round(frac_hbm, 5),
            "achieved_tflops": round(alg_flops / launch_s / 1e12, 3), "achieved_gbs": round(alg_bytes / launch_s / 1e9, 1),
            "note": "bound chosen by algorithmic intensity (flops / algorithmic bytes) vs the ridge peak_flops / 8 TB/s; "
                    "achieved = algorithmic flops (2 N^2 B C per support and hop, true channel count) or algorithmic bytes "
                    "(supports + input plane + propagated planes, once each) / HIP-event launch time; traffic = corrected "
                    "PMC HBM bytes per launch (profiles/, tools/pmc_traffic.sh)"
                    + ("; bf16x3 issues 3 bf16 MFMAs per product: its matrix-core ceiling is 833 TF" if dtype == "bf16x3" else ""),
            "avg_launch_us": round(1e3 * ms / n, 3), "launches": n,
            "alg_flops_per_launch": alg_flops, "alg_bytes_per_launch": alg_bytes,
            # the clock the chip HELD inside these launches' K loops (in-kernel stamps of workgroup 0: shader cycles / 100 MHz wall clock).
            # `peak` above is the dense peak at 2.4 GHz; a power-limited part holds less under a dense-MFMA loop, and that is not the kernel's to win
            **({"shader_clock_mhz": mhz, "clock_stamped_launches": n_stamped,
                "frac_of_mfma_peak_at_held_clock": round(frac_mfma * MAX_CLOCK_MHZ / mhz, 5)} if mhz else {})}


TILE_DIR = os.path.join(ROOT, "profiles", "tiles")


def tile_cache_path(config_name, B, prec):
    return os.path.join(TILE_DIR, f"{config_name}_B{B}_{prec}.json")


def tile_cache_current(config_name, B, prec):
    """a committed tile table of this shape exists AND was written by this build of the library"""
    from megacrn_amd import _lib
    path = tile_cache_path(config_name, B, prec)
    try:
        return os.path.exists(path) and json.load(open(path)).get("build_id") == _lib.build_id()
    except (OSError, ValueError):
        return False


def import_tile_cache(config_name, B, prec):
    """GEMM tile table of this (config, batch, arithmetic) from an earlier mcrn_model_autotune on an MI355X (committed under
    profiles/tiles/ by `--save-tiles`): with it the one-off tuning of the N = 8192 shape (about a minute) costs nothing, so the SYN
    leg fits the default run.  Tiles only select among equivalent kernels (speed and fp32 summation order, never results beyond
    that); a table of another library version is ignored and the shape is tuned as usual."""
    from megacrn_amd import _lib
    path = tile_cache_path(config_name, B, prec)
    if not os.path.exists(path):
        return False
    rec = json.load(open(path))
    if rec.get("build_id") != _lib.build_id():        # (tile slots and their kernels belong to a build: a hand-bumped version number went stale in round 5)
        return False
    try:
        _lib.autotune_import(rec["words"])
    except RuntimeError:
        return False
    return True


def make_trainer(config_name, B, prec, device, rank, tile_cache=False, save_tiles=None):
    import megacrn_amd
    from megacrn_amd import _lib
    from megacrn_amd.trainer import FlatTrainer
    cfg = CONFIGS[config_name]
    model = megacrn_amd.MegaCRN(num_nodes=cfg["N"], input_dim=1, output_dim=1, horizon=cfg["T"],
                                rnn_units=cfg["H"], mem_num=cfg["M"], mem_dim=cfg["D"]).to(device).train()
    model.precision = _lib.PRECISIONS[prec]
    tr = FlatTrainer(model, lr=0.01, eps=1e-3, max_grad_norm=5, scaler_mean=SC_MEAN, scaler_std=SC_STD)
    batch = synth(cfg, B, 1234 + rank, device)
    tr.tiles_cached = bool(tile_cache) and import_tile_cache(config_name, B, prec)
    t_prep = time.perf_counter()
    tr._prepare(batch[0])              # workspace + one-off GEMM tile autotune (only signatures the table lacks): never inside a timed region
    torch.cuda.synchronize()
    tr.autotune_s = time.perf_counter() - t_prep      # (+ dp.share_autotune's broadcast at world > 1: DESIGN.md section 6)
    if save_tiles:
        os.makedirs(os.path.dirname(os.path.abspath(save_tiles)), exist_ok=True)
        json.dump({"config": config_name, "B": B, "precision": prec, "lib_version": _lib.lib.mcrn_version(), "build_id": _lib.build_id(),
                   "words": [int(w) for w in _lib.autotune_export()]}, open(save_tiles, "w"))
    return tr, batch


LEG_INDEX = {"expytky": 3, "syn8192": 4}


def leg_workspace_bytes(name, prec):
    """what mcrn_model_workspace_bytes asks for the leg's shape (the library sizes for 288 GB parts: every step's planes are kept)"""
    from megacrn_amd import _lib
    cfg = CONFIGS[name]
    d = _lib.Dims(cfg["B"], cfg["N"], cfg["T"], cfg["T"], 1, 1, 1, cfg["H"], cfg["M"], cfg["D"], 3, _lib.PRECISIONS[prec])
    return int(_lib.lib.mcrn_model_workspace_bytes(C.byref(d)))


def secondary_leg(device, steps=10, warmup=3, name="expytky", regimes=True, prec="bf16", nrep=4, tile_cache=False):
    """The north_star figure, driver-timed: forward N x N propagation at N = 1843 (EXPY-TKY shape, B = 32, T = 6, H = 32) as a
    short extra run after the headline measurement (a few seconds), in the arithmetic `prec`:
      bf16   - the bf16-resident large-graph mode (one MFMA per product; stated tolerance 1e-2, tests/test_gpu_parity.py::test_bf16_mode_*)
      bf16x3 - the parity arithmetic (3 bf16 MFMAs per product, fp32 storage; 1e-4 as north_star states): `secondary_parity`
    name = "syn8192": the same leg on BASELINE configs[4], the N = 8192 roofline run (tile table from profiles/tiles/)."""
    cfg = CONFIGS[name]
    B = cfg["B"]
    tr, batch = make_trainer(name, B, prec, device, 0, tile_cache=tile_cache)
    for _ in range(warmup):
        tr.train_step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(*batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    regimes = regime_legs(tr, batch, B, steps, 2, torch.cuda.synchronize) if regimes else {}
    roof = roofline_of(tr, batch, cfg, name, B, prec, nrep=nrep)
    ms = 1e3 * dt / steps
    tol = {"bf16": "1e-2 (bf16-resident propagation operands; measured <= 5.4e-3)", "bf16x3": "1e-4 (the tolerance north_star states)",
           "f32": "1e-4"}[prec]
    return {**regimes, "what": f"forward K-hop propagation at N={cfg['N']} (BASELINE configs[{LEG_INDEX[name]}] shape, per-GPU batch {B}), "
                    "the kernel north_star sets the >= 40 % bf16-MFMA target on; measured in this same process after the headline run",
            "config": {"workload": f"{cfg['label']} N={cfg['N']} T_in=T_out={cfg['T']} rnn_units={cfg['H']} mem={cfg['M']}x{cfg['D']} "
                                   f"cheb_k=3, batch {B}, full train step"},
            "dtype": prec, "parity_tolerance": tol, "value": round(B * steps / dt, 2), "unit": "samples/s", "ms_per_step": round(ms, 4),
            "steps": steps, "warmup": warmup, "tile_table_cached": bool(getattr(tr, "tiles_cached", False)),
            "autotune_s": round(tr.autotune_s, 2),
            **step_traffic(name, prec, ms), "roofline": roof}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="metrla", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU runs the config batch; strong: the config batch is split over the GPUs")
    ap.add_argument("--precision", default=None, choices=["f32", "bf16x3", "bf16"], help="default: per config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the N=1843 propagation leg of the default run")
    ap.add_argument("--batches-seen", type=int, default=0,
                    help="curriculum position of the first timed step (model/MegaCRN.py:146-147): 0 = every step teacher-forced "
                         "(the start of training); the line also carries value_no_teacher (no step teacher-forced) either way")
    ap.add_argument("--no-regimes", action="store_true", help="skip the no-teacher and evaluation-forward legs")
    ap.add_argument("--with-syn", action="store_true",
                    help="force the third leg (3 train steps + forward-propagation roofline of the N = 8192 stress shape, BASELINE "
                         "configs[4]; ~35 GB of workspace): by default it runs with 2 steps when profiles/tiles/ holds the tile table of "
                         "the shape (otherwise its one-off tuning takes about a minute) and its ~85 GB fit the free HBM")
    ap.add_argument("--no-syn", action="store_true", help="skip the N = 8192 leg of the default run")
    ap.add_argument("--save-tiles", default=None, help="write the GEMM tile table of this run's shape to PATH after tuning "
                                                       "(commit it as profiles/tiles/<config>_B<batch>_<precision>.json)")
    ap.add_argument("--roles", default="1,2,3,4,5,6,7", help="GEMM roles timed for gemm_roles (diagnostics); 7 = the hoisted once-per-stack\n                    input-channel products of the bf16 mode (absent in the other modes)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import megacrn_amd  # noqa: F401
    from megacrn_amd import dp
    from megacrn_amd._lib import lib
    import torch.distributed as dist

    rank, local_rank, env_world = dp.init_from_env("nccl")
    # what the process group itself reports - not the environment - plus an all-reduce of ones below: the line proves
    # that RCCL connected `world` ranks
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world != args.gpus or env_world != world:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {world} ranks (WORLD_SIZE={env_world}): "
                         f"launch one rank per GPU (python bench.py --gpus N does)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, device=device)
        dist.all_reduce(one)                       # RCCL all-reduce(sum) of ones = number of ranks that took part
        ranks_seen = int(round(one.item()))
    cfg = CONFIGS[args.config]
    B = args.batch or cfg["B"]
    if args.scaling == "strong":
        if B % world:
            raise SystemExit(f"strong scaling: batch {B} is not divisible by {world} GPUs")
        B //= world

    torch.manual_seed(1234)            # identical init on every rank (also broadcast by the trainer)
    dp.seed_curriculum(1234)           # shared numpy stream: same teacher-forcing draws on all ranks
    # arithmetic: bf16x3 (fp32-equivalent, the 1e-4 parity mode) for the small graphs; the large graphs default to the
    # bf16-resident propagation mode (own stated tolerance, tests/test_gpu_parity.py::test_bf16_mode_*)
    prec = args.precision or ("bf16" if cfg["N"] >= 1024 else "bf16x3")
    # (a committed tile table of exactly this config / batch / arithmetic / library version is adopted - today only N = 8192 has one -
    #  unless this run is the one that writes it)
    tr, batch = make_trainer(args.config, B, prec, device, rank, tile_cache=not args.save_tiles,
                             save_tiles=args.save_tiles if rank == 0 else None)
    autotune_s = tr.autotune_s
    x, ycov, y = batch
    dtype = prec

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the step's one collective, alone: all-reduce(sum) of a buffer of the flat gradient bucket's size over RCCL (N > 1)
    allreduce_us, bucket_bytes = None, int(tr.flat_g.numel() * 4)
    if world > 1:
        probe = torch.zeros_like(tr.flat_g)
        for _ in range(5):
            dist.all_reduce(probe)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(probe)
        torch.cuda.synchronize()
        allreduce_us = round(1e6 * (time.perf_counter() - t0) / 20, 2)
        del probe
    tr.batches_seen = args.batches_seen
    loss = torch.zeros((), device=device)
    for _ in range(args.warmup):
        loss = tr.train_step(x, ycov, y)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.train_step(x, ycov, y)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    final_loss = float(loss.item())
    launches = lib.mcrn_last_launch_count()
    regimes = None
    if not args.no_regimes:
        regimes = regime_legs(tr, batch, B * world, max(4, min(args.steps, 20)), 2, sync_all)

    # ---- roofline leg: HIP events around every launch of one GEMM role during real train steps
    roles, roof = {}, None
    if not args.no_roofline:
        for role in [int(r) for r in args.roles.split(",") if r]:
            ms, n, af, ef = time_role(tr, batch, role, 1)
            if n == 0:
                continue
            roles[ROLE_NAMES[role]] = dict(ms_per_step=round(ms, 4), launches_per_step=n,
                                           avg_us=round(1e3 * ms / max(n, 1), 3),
                                           alg_tflops=round(af / (ms * 1e-3) / 1e12, 2) if ms else 0,
                                           exec_tflops=round(ef / (ms * 1e-3) / 1e12, 2) if ms else 0)
        sync_all()
        # dominant kernel of the path = forward K-hop propagation (north_star): average over several steps
        roof = roofline_of(tr, batch, cfg, args.config, B, dtype)
        # the kernel that takes the most time per step, when it is not the propagation
        if roles:
            top = max(roles.items(), key=lambda kv: kv[1]["ms_per_step"])
            # its own roofline fractions: algorithmic flops / launch time against the dense MFMA peak of the arithmetic (bf16x3 issues 3
            # bf16 MFMAs per product: the fraction of the 833 TF it can reach is 3 x this one)
            roof["largest_role_by_time"] = {"role": top[0], **top[1],
                                            "frac_of_mfma_peak": round(top[1]["alg_tflops"] * 1e12 / PEAK[dtype], 5)}
        sync_all()

    secondary = secondary_parity = None
    if rank == 0 and world == 1 and args.config == "metrla" and not args.no_secondary and not args.no_roofline:
        del tr
        torch.cuda.empty_cache()
        try:
            secondary = secondary_leg(device)
        except Exception as e:                      # the headline line must not depend on the extra leg
            secondary = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()
        try:   # the same shape in the arithmetic that meets north_star's 1e-4 (3 MFMAs per product: its matrix-core ceiling is 833 TF)
            secondary_parity = secondary_leg(device, steps=8, warmup=2, prec="bf16x3", regimes=False, nrep=2)
        except Exception as e:
            secondary_parity = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()

    tertiary = None
    # N = 8192 (BASELINE configs[4]): default when the tile table of the shape is committed (no minute of tuning) and the leg's workspace (~80 GB) fits the free HBM
    syn_auto = (args.config == "metrla" and not args.no_secondary and not args.no_roofline and not args.no_syn
                and tile_cache_current("syn8192", CONFIGS["syn8192"]["B"], "bf16"))
    if rank == 0 and world == 1 and (args.with_syn or syn_auto):
        torch.cuda.empty_cache()
        free_b, total_b = torch.cuda.mem_get_info()
        need_b = leg_workspace_bytes("syn8192", "bf16") * 1.05 + 4e9        # workspace + batch, outputs, gradient views, allocator slack
        # (unasked, the leg takes at most HALF of the part's HBM and only memory that is free now: a default run must not crowd a co-tenant out)
        if (free_b < need_b or need_b > 0.5 * total_b) and not args.with_syn:
            tertiary = {"skipped": f"{free_b / 1e9:.0f} of {total_b / 1e9:.0f} GB of HBM free, the N = 8192 leg needs ~{need_b / 1e9:.0f} GB "
                                   f"(run by default only when that is free and at most half of the part; --with-syn forces it)"}
        else:
            try:
                tertiary = secondary_leg(device, steps=3, warmup=1, name="syn8192", regimes=False, nrep=1,
                                         tile_cache=True)
            except Exception as e:
                tertiary = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded sample (about 10-30 s of CPU work in total).  Main figure: the reference's own ATen op sequence on the host
        # (oracle/megacrn_torch_cpu.py) at min(cores, 32) threads and at ONE thread (what the reference trainer pins,
        # model/traintest_MegaCRN.py:255-261): the full batch when a step fits ~1e12 algorithmic flops per leg, else as many
        # samples / sequence steps as fit (scaled linearly).  Second figure: the numpy port (the parity oracle) on one thread.
        # (the best of 8 / 16 / 32 threads: the op sizes of the small graphs do not scale to 32)
        sB, sT = cpu_sample(cfg, B, 1.4e12)
        sweep = {}
        for cand in sorted({min(c, os.cpu_count() or 1) for c in (8, 16, 32)}):
            sweep[cand] = cpu_baseline_torch(cfg, sB, sT, cand)
        nthr = max(sweep, key=lambda c: sweep[c][0])
        v, secs = sweep[nthr]
        s1, t1 = cpu_sample(cfg, B, 0.7e12)
        v1, secs1 = cpu_baseline_torch(cfg, s1, t1, 1)
        sn, tn = cpu_sample(cfg, B, 0.25e12)
        vn, secsn = cpu_baseline_numpy(cfg, sn, tn, 1)
        cpu = {"value": round(v, 4), "unit": "samples/s", "cores": nthr, "kind": "port",
               "value_1thread": round(v1, 4), "value_numpy_port_1thread": round(vn, 4),
               "thread_sweep": {str(c): round(sweep[c][0], 4) for c in sorted(sweep)},
               "cpu_model": cpu_model_string(), "host_cores": os.cpu_count(),
               "sample": f"oracle/megacrn_torch_cpu.py (the reference's ATen op sequence on PyTorch-CPU, autograd backward, "
                         f"clip_grad_norm_, torch Adam), warm (one untimed step on 2 samples first), one full train step of the "
                         f"same {cfg['label']} workload: {sB} of the {B} samples x {sT} of {cfg['T']} sequence steps on {nthr} "
                         f"threads (the best of {sorted(sweep)}: thread_sweep; {secs:.1f} s); value_1thread: {s1} samples x {t1} steps on 1 thread ({secs1:.1f} s); "
                         f"value_numpy_port_1thread: oracle/megacrn_oracle.py, {sn} samples x {tn} steps ({secsn:.1f} s); "
                         f"all scaled linearly to the full sequence length"}

    if rank == 0:
        gb = B * world
        val = gb * args.steps / dt
        step_flops = 3.0 * alg_flops_forward(cfg, B)
        out = {
            "metric": "training samples/sec (12-step seq2seq)" if cfg["T"] == 12 else "training samples/sec (6-step seq2seq)",
            "value": round(val, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": args.scaling,
            "backend": "rccl" if world > 1 else "single-process", "world_size": world, "rccl_ranks_seen": ranks_seen,
            "allreduce_us": allreduce_us, "allreduce_bytes": bucket_bytes,
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{cfg['label']} N={cfg['N']} T_in=T_out={cfg['T']} rnn_units={cfg['H']} "
                                   f"mem={cfg['M']}x{cfg['D']} cheb_k=3, per-GPU batch {B}, full train step "
                                   f"(fwd + 3-term loss + bwd + all-reduce + clip + Adam)",
                       "global_batch": gb, "parallelism": f"dp{world}"},
            "step_alg_tflops": round(step_flops * world / (dt / args.steps) / 1e12, 2),
            "kernel_launches_per_step": launches, "final_loss": round(final_loss, 5),
            "autotune_s": round(autotune_s, 2),      # workspace + one-off tile autotune (+ tile-table broadcast), before the timed region
            "teacher_regime": f"timed steps start at batches_seen = {args.batches_seen}: teacher-forcing probability "
                              f"{2000.0 / (2000.0 + float(np.exp(min(args.batches_seen, 10 ** 6) / 2000.0))):.3f} "
                              f"(cl_decay_steps 2000, model/MegaCRN.py:146-147); value_no_teacher = no step teacher-forced",
        }
        if regimes:
            out.update(regimes)
        if roof:
            out["roofline"] = roof
            out["gemm_roles"] = roles
        out.update(step_traffic(args.config, dtype, out["ms_per_step"]))
        if secondary:
            out["secondary"] = secondary
        if secondary_parity:
            out["secondary_parity"] = secondary_parity
        if tertiary:
            out["syn8192"] = tertiary
        if cpu:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
