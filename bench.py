#!/usr/bin/env python3
"""bench.py -- HMC leapfrog steps/s on a 256^3 grid (BASELINE.json metric), one chain per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one leapfrog step (HMC.cc:284-365 body: half kick, M^-1 p, drift, one full force evaluation,
half kick).  The timed region is ONE trajectory of exactly K steps through the C ABI
(bchmc_leapfrog_device: state transforms + initial force + K steps + inverse transforms), with q0/p0 and all
input grids already resident in HBM, bracketed by barrier + torch.cuda.synchronize().  Each rank runs its own
independent chain (weak scaling); the only collective is the epsilon-statistics all-gather (one 520-byte packet
per rank per sample, bchmc_eps_exchange; RCCL).  Rank 0 prints one JSON line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N ranks: the parent
process starts N children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set) before it
touches the GPU -- it never does -- waits for them, relays rank 0's JSON line and propagates a non-zero exit code.
Under torch.distributed.run the ranks exist already; --gpus must then equal WORLD_SIZE.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_CELL_STEP = 544  # 68 real arrays x 8 B per leapfrog step (default force path)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nx", type=int, default=256, help="grid cells per axis (BASELINE config 3: 256)")
    ap.add_argument("--likelihood", type=int, default=1)
    ap.add_argument("--no-rsd", action="store_true")
    ap.add_argument("--alpt", action="store_true",
                    help="with --no-rsd: sfmodel = 2, the ALPT forward model (Lag2Eul_non_zeldovich) in every force evaluation")
    ap.add_argument("--mk", type=int, default=3, help="masskernel 0 NGP / 1 CIC / 2 TSC / 3 SPH (default; anything else is a "
                                                      "separate line, never the headline)")
    ap.add_argument("--calc-h", type=int, default=2, help="likelihood-force variant 0..3 (default 2; mk != 3 needs 0 or 1)")
    ap.add_argument("--sustained", type=float, default=3.0,
                    help="N = 1: additionally run ONE trajectory long enough for about this many seconds and report it as "
                         "`sustained` next to the headline (untimed part of the run; 0 switches it off)")
    ap.add_argument("--no-rccl-probe", action="store_true",
                    help="N > 1 on nccl: skip the untimed ncclAllGather through the library's own RCCL transport")
    ap.add_argument("--force-rccl-probe", action="store_true",
                    help="rehearsal: run that probe whatever the backend (two ranks on one device make RCCL refuse: the "
                         "error path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-nx", type=int, default=0, help="grid of the CPU-baseline sample (default: same as --nx)")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--fp32", action="store_true", help="fp32 field arrays (BASELINE config 5); tolerance re-stated")
    ap.add_argument("--chains-per-gpu", type=int, default=1, choices=[1, 2],
                    help="2: additionally measure the throughput mode -- two chains per GPU on disjoint halves of the "
                         "CUs -- and report it as `throughput_mode` next to the headline (never instead of it)")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--exchange", default="torch", choices=["torch", "rccl"],
                    help="transport of the per-sample epsilon-record exchange behind bchmc_eps_exchange: torch.distributed "
                         "all_gather (default) or the library's own ncclAllGather (unique id broadcast through torch)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (never for reported numbers)")
    return ap.parse_args()


def _profile_key(path):
    """Sort key of a profiles/ file name: (round, version) as NUMBERS -- r02_v9 < r02_v17 < r03_v1 -- then the name."""
    import re
    name = os.path.basename(path)
    m = re.match(r"r(\d+)(?:_v(\d+))?", name)
    return (int(m.group(1)) if m else -1, int(m.group(2)) if (m and m.group(2)) else 0, name)


def pmc_traffic(nx, precision):
    """HBM bytes per leapfrog step from the newest committed rocprofv3 PMC summary (profiles/r*_pmc_traffic.json,
    made by scripts/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE passes of this bench command).
    Returns None when no summary exists for this grid."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), key=_profile_key):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("grid", 256) == nx and d.get("precision", "fp64") == precision:
            best = (path, d)
    if best is None:
        return None, None
    return best[1]["hbm_bytes_per_step"], os.path.relpath(best[0], ROOT)


def _rel_l2(a, b):
    import numpy as np
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def cpu_baseline(params, case_arrays, q0, p0, eps, reps=3):
    """Oracle (kind "port") timed on this box's host cores on a bounded sample of the same workload
    (SURVEY 8d protocol): OpenMP build on all host cores, `reps` repetitions, per-step time of a repetition =
    (t[3-step trajectory] - t[1-step trajectory]) / 2 (the initial force evaluation and the state copies cancel),
    median reported; plus a 1-core figure from the serial build on a 64^3 sample of the same recipe.
    Returns (json object, (q1, p1) of the oracle's 3-step trajectory for the parity-at-size check)."""
    import numpy as np
    from oracle.oracle import Oracle
    o = Oracle(params, omp=True)
    cores = int(o.lib.orc_get_max_threads())
    o.set(**case_arrays)
    per_step, total, traj3 = [], 0.0, None
    for _ in range(reps):
        t0 = time.perf_counter()
        o.Hamiltonian_EoM(q0, p0, eps, 1)
        t1 = time.perf_counter()
        q1, p1, _done = o.Hamiltonian_EoM(q0, p0, eps, 3)
        t2 = time.perf_counter()
        per_step.append(((t2 - t1) - (t1 - t0)) / 2.0)
        total += t2 - t0
        if traj3 is None:
            traj3 = (np.array(q1, copy=True), np.array(p1, copy=True))
    o.close()
    med = float(np.median(per_step))
    # 1 core: the serial build (same source, gcc -O2) on a 64^3 grid of the same recipe, 3 repetitions
    from tests.util import Case
    c1 = Case(Nx=64, L=params.L * 64.0 / params.Nx, likelihood=params.likelihood, rsd_model=params.rsd_model,
              sfmodel=params.sfmodel)
    one = []
    for _ in range(reps):
        t0 = time.perf_counter()
        c1.oracle.Hamiltonian_EoM(c1.q0, c1.p0, c1.eps, 1)
        t1 = time.perf_counter()
        c1.oracle.Hamiltonian_EoM(c1.q0, c1.p0, c1.eps, 3)
        t2 = time.perf_counter()
        one.append(((t2 - t1) - (t1 - t0)) / 2.0)
        total += t2 - t0
    c1.oracle.close()
    one_med = float(np.median(one))
    return dict(value=1.0 / med, unit="steps/s", cores=cores, kind="port", repetitions=reps,
                per_step_s=[round(x, 3) for x in per_step],
                cell_steps_per_s=params.N / med,
                one_core=dict(value=1.0 / one_med, unit="steps/s", grid=64, cores=1,
                              cell_steps_per_s=64 ** 3 / one_med, per_step_s=[round(x, 4) for x in one],
                              build="oracle/liboracle.so (serial, gcc -O2)"),
                fft_backend="oracle/orc_fft.c (bundled radix-2/4 + Bluestein row FFTs, OpenMP over rows; no FFTW3 on "
                            "this image)",
                sample="oracle/liboracle_omp.so (gcc -O3 -march=x86-64-v3 -fopenmp, %d threads), %d^3 grid, same "
                       "inputs as the GPU run: median over %d repetitions of (t[3-step trajectory] - t[1-step "
                       "trajectory])/2 = %.2f s per leapfrog step; 1-core figure: serial build, 64^3 grid of the same "
                       "recipe, %.3f s per step; %.0f s of CPU work in total" % (cores, params.Nx, reps, med, one_med,
                                                                               total)), traj3


def valu_roofline():
    """fp64 vector-ALU roofline of the two particle-mesh kernels from the newest committed SQ-counter summary
    (profiles/r*_sq_tile81.json, made by scripts/pmc_sq.py from separate rocprofv3 --pmc passes of this bench
    command; PMC needs its own passes, so this is never measured inside the timed run)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_tile81.json")), key=_profile_key)
    if not paths:
        return None
    try:
        d = json.load(open(paths[-1]))
    except (OSError, ValueError):
        return None
    d["source"] = os.path.relpath(paths[-1], ROOT)
    d["measured_in_run"] = False
    return d


def two_chain_throughput(params, fields, arrays, dev, local_rank, eps, steps, precision):
    """Throughput mode: two independent chains on ONE GPU, each handle's stream restricted to half of the CUs
    (BCHMC_CU_MASK=lo / hi, hipExtStreamCreateWithCUMask), so that the VALU-bound particle-mesh kernels of one chain
    run beside the HBM-bound transforms of the other.  Aggregate steps/s of both chains; per-chain latency doubles."""
    import torch
    from barcode_amd import inputs
    from barcode_amd.engine import Engine
    engines, states = [], []
    for c, mask in enumerate(("lo", "hi")):
        os.environ["BCHMC_CU_MASK"] = mask
        e = Engine(params, device=local_rank, precision=precision)
        e.upload(**arrays)
        p0 = torch.from_numpy(inputs.gaussian_random_field(params, fields["mass_f"], inputs.SEED_P0 + 100 + c).reshape(-1)).to(dev)
        q0 = torch.from_numpy(fields["q0"].reshape(-1)).to(dev)
        engines.append(e)
        states.append((q0, p0, torch.empty_like(q0), torch.empty_like(p0)))
    os.environ.pop("BCHMC_CU_MASK", None)
    for e, st in zip(engines, states):
        e.leapfrog_device(*st, eps, 3)
    for e in engines:
        e.sync()
    t0 = time.perf_counter()
    for e, st in zip(engines, states):   # both trajectories are enqueued asynchronously, each on its own stream
        e.leapfrog_device(*st, eps, steps)
    done = [e.steps_done() for e in engines]   # synchronises each stream
    dt = time.perf_counter() - t0
    for e in engines:
        e.close()
    return dict(chains_per_gpu=2, value=round(2 * steps / dt, 4), unit="steps/s (aggregate of both chains)",
                ms_per_step_per_chain=round(1e3 * dt / steps, 4), steps_done=[int(d) for d in done],
                cu_masks=["lo", "hi"],
                note="opt-in deployment mode for many-chain sampling; the headline `value` is one chain on the whole GPU")


def rccl_native_probe(dist, dev, rank, world, eps, steps, timeout_s=90.0):
    """Untimed: the same exchange once more through the LIBRARY's own RCCL transport (bchmc_comm_create: ncclCommInitRank
    + ncclAllGather behind the C ABI, what a barcode/main.cc-driven chain uses), whatever transport the timed exchange
    ran on.  Isolated from the measurement: the 128-byte id travels over a gloo side group (so nothing of the probe is
    queued on torch's RCCL process group), the library's communicator is its own, and the work runs in a thread with a
    deadline -- a transport that cannot come up is reported (`hung`), not waited for."""
    import threading
    from barcode_amd import engine as _eng
    res = dict(ok=False)
    side = dist.new_group(backend="gloo")  # collective: every rank, main thread

    def work():
        try:
            uid = [_eng.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0, group=side)
            c = _eng.Comm(rank, world, device=dev.index, unique_id=uid[0])
            got = c.exchange([(eps, True, steps)])
            info = c.info()
            res.update(transport=info["transport"], world_seen=info["world"],
                       ranks_seen=sorted({int(r[0]) for r in got}), records=len(got))
            res["ok"] = bool(info["transport"] == "rccl" and info["world"] == world and
                             res["ranks_seen"] == list(range(world)) and len(got) == world)
            c.close()
        except Exception as e:  # reported in the line, never fatal for the measurement
            res["error"] = "%s: %s" % (type(e).__name__, e)

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        res["error"] = "no answer within %.0f s" % timeout_s
        res["hung"] = True
    return res


def sustained_run(engine, q0, p0, q1, p1, eps, ms_per_step, seconds, torch):
    """ONE trajectory of about `seconds` seconds (steps from the headline's ms per step), timed on the engine's stream,
    with a split per fifth of the trajectory taken from HIP events recorded by a second, shorter pass."""
    steps = max(int(seconds * 1e3 / ms_per_step), 10)
    t0 = time.perf_counter()
    engine.leapfrog_device(q0, p0, q1, p1, eps, steps)
    done = engine.steps_done()
    dt = time.perf_counter() - t0
    # per-segment rates: five consecutive trajectories of steps / 5 (each carries the ~1.1-step fixed cost)
    seg, rates = max(steps // 5, 1), []
    for _ in range(5):
        s0 = time.perf_counter()
        engine.leapfrog_device(q0, p0, q1, p1, eps, seg)
        engine.steps_done()
        rates.append(round(seg / (time.perf_counter() - s0), 2))
    return dict(value=round(steps / dt, 4), unit="steps/s", steps=steps, seconds=round(dt, 3), steps_done=int(done),
                ms_per_step=round(1e3 * dt / steps, 4), fifths_steps_per_s=rates,
                note="one trajectory, inputs resident, wall clock around launch + synchronise")


def launch_ranks(args):
    """Parent of a self-launched multi-GPU run: start one child per GPU, never touch the GPU here."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()  # counting devices does not initialise the GPU runtime in this process
    # the children are fresh processes (never a re-exec of this one), and this one must stay off the GPU for good
    assert not torch.cuda.is_initialized(), "bench.py parent initialised the GPU before spawning its ranks"
    if args.gpus > ndev and not args.single_device:
        print("bench.py: --gpus %d but this node shows %d GPU(s): refusing to run" % (args.gpus, ndev), file=sys.stderr)
        sys.exit(2)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   BCHMC_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # wait for all ranks; a rank that dies (no such device, out of memory, ...) leaves the others blocked in their
    # rendezvous or barrier: end them (our own children, by PID) instead of waiting for a collective timeout
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = pr.poll()
        if any(rc not in (None, 0) for rc in rcs):
            time.sleep(2.0)  # let the failing rank's message reach stderr first
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    pr.kill()
                    rcs[i] = pr.wait()
            break
        time.sleep(0.1)
    assert not torch.cuda.is_initialized()
    out0 = procs[0].stdout.read()  # one JSON line: far below the pipe buffer, safe to read after the exit
    # exactly one JSON line on stdout: libraries of the children may have written their own chatter there
    for line in out0.decode().splitlines():
        if line.startswith("{"):
            print(line)
        elif line.strip():
            print(line, file=sys.stderr)
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc]
    if bad:
        print("bench.py: rank exit codes %s" % rcs, file=sys.stderr)
        sys.exit(bad[0] if bad[0] > 0 else 1)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)  # parent: no torch.cuda / engine call has happened or will happen here
    # host threads for the CPU baseline and the input generation: the CPU share of this process (affinity capped by
    # the cgroup quota), not the visible core count; must be set before the OpenMP library loads
    from barcode_amd.inputs import host_cpu_share
    ncores = host_cpu_share()
    os.environ.setdefault("OMP_NUM_THREADS", str(ncores))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    if args.gpus != world:
        # a wrong rank count must never be recorded as an N-GPU number
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d: refusing to run" % (args.gpus, world), file=sys.stderr)
        if distributed:
            dist.destroy_process_group()
        sys.exit(2)
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from barcode_amd import inputs
    from barcode_amd.chains import ChainGroup, EpsRing
    from barcode_amd.engine import Engine
    from barcode_amd.params import HamilParams

    rsd = 0 if args.no_rsd else 1
    # BASELINE config 3: "256^3, 2LPT + RSD": under rsd_model the reference dispatches to Zel'dovich + plane-parallel
    # RSD whatever sfmodel says; Gaussian likelihood, SPH kernel, calc_h 2, mass_type 1, fp64.
    params = HamilParams(Nx=args.nx, L=200.0, likelihood=args.likelihood, rsd_model=rsd, sfmodel=2 if (rsd or args.alpt) else 1,
                         mk=args.mk, calc_h=args.calc_h)
    headline_variant = (args.mk == 3 and args.calc_h == 2)
    group = ChainGroup(pool=True, device=dev if args.backend == "nccl" else torch.device("cpu"),
                       transport=args.exchange)
    ring = EpsRing()

    f = inputs.make_fields(params)
    engine = Engine(params, device=local_rank, precision=1 if args.fp32 else 0)
    engine.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"])
    # mock data: forward model of the truth field on the GPU, then the reference's noise model
    if params.likelihood != 3:
        engine.upload(nobs=np.zeros(params.N), window=np.ones(params.N), noise=np.ones(params.N))
        engine.forward(f["truth"], rsd if params.likelihood == 1 else 0)
        dX = engine.fetch("deltaX").reshape((params.Nx,) * 3)
    else:
        dX = np.zeros((params.Nx,) * 3)
    window, noise, nobs = inputs.mock_observations(params, dX, delta_lag=f["truth"])
    engine.upload(window=window, noise=noise, nobs=nobs)
    arrays = dict(signal_PS=f["signal_PS"], mass_f=f["mass_f"], window=window, noise=noise, nobs=nobs)

    # independent chains: same data, different momenta per rank (seed + rank)
    p0_host = inputs.gaussian_random_field(params, f["mass_f"], group.chain_seed(inputs.SEED_P0))
    q0 = torch.from_numpy(f["q0"].reshape(-1)).to(dev)
    p0 = torch.from_numpy(p0_host.reshape(-1)).to(dev)
    q1, p1 = torch.empty_like(q0), torch.empty_like(p0)
    eps = 0.5 * params.eps_heuristic()  # 0.5 * 2.38902581 * N^-0.57495347, fixed

    def barrier():
        if distributed:
            dist.barrier()

    # ---- warmup (untimed) -------------------------------------------------------------------------
    if args.warmup > 0:
        engine.leapfrog_device(q0, p0, q1, p1, eps, args.warmup)
        engine.sync()
    got_w = group.pool_into(ring, [(eps, True, args.warmup)])  # untimed: first use of the collective sets up RCCL's channels
    stream = torch.cuda.ExternalStream(engine.stream, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    # ---- timed: exactly K leapfrog steps -----------------------------------------------------------
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    engine.leapfrog_device(q0, p0, q1, p1, eps, args.steps)
    ev1.record(stream)
    done = engine.steps_done()  # synchronises the engine's stream
    ring.record(True, eps)
    got_t = group.pool_into(ring, [(eps, True, args.steps)])  # the path's only collective: one 520-byte packet per rank per sample
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    gpu_ms = ev0.elapsed_time(ev1)
    if distributed:
        t = torch.tensor([wall], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    # what the collective itself saw (so that a recorded N-GPU line needs no inference about the rank count): the
    # communicator's own world size, the set of ranks whose records arrived in the timed exchange, the record count
    collective = dict(backend=(args.backend if distributed else None),
                      transport=(group.comm.info()["transport"] if group.comm is not None else "none"),
                      exchange=args.exchange if distributed else None,
                      world_seen=(group.comm.info()["world"] if group.comm is not None else 1),
                      ranks_seen=sorted({int(r[0]) for r in got_t}), records=len(got_t),
                      warmup_records=len(got_w), in_timed_region=True)
    if distributed:
        # every rank must have seen the same thing
        seen = torch.tensor([collective["world_seen"], len(collective["ranks_seen"]), collective["records"]],
                            dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        lo, hi = seen.clone(), seen.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        collective["consistent_across_ranks"] = bool(torch.equal(lo, hi))
        collective["ok"] = bool(collective["consistent_across_ranks"] and collective["world_seen"] == world and
                                collective["ranks_seen"] == list(range(world)) and collective["records"] == world)
    else:
        collective["ok"] = collective["ranks_seen"] == [0] and collective["records"] == 1
    if distributed and ((args.backend == "nccl" and not args.single_device and not args.no_rccl_probe) or
                        args.force_rccl_probe):
        try:
            collective["rccl_native"] = rccl_native_probe(dist, dev, rank, world, eps, args.steps,
                                                          timeout_s=20.0 if args.force_rccl_probe else 90.0)
        except Exception as e:  # the probe must never cost the measurement
            collective["rccl_native"] = dict(ok=False, error="%s: %s" % (type(e).__name__, e))
    if done != args.steps:
        print("bench.py: runaway guard fired after %d of %d steps" % (done, args.steps), file=sys.stderr)
    finite = bool(torch.isfinite(q1).all().item() and torch.isfinite(p1).all().item())

    # ---- per-kernel HIP-event breakdown (separate, untimed pass) -------------------------------------
    kernels = dominant = None
    if not args.no_kernel_profile:
        ksteps = min(args.steps, 10)
        engine.profile(True)
        engine.leapfrog_device(q0, p0, q1, p1, eps, ksteps)
        engine.sync()
        prof = engine.profile_read()
        engine.profile(False)
        kernels = {k: dict(ms_per_step=round(ms / ksteps, 4), launches=n, avg_launch_ms=round(ms / max(n, 1), 4))
                   for k, (ms, n) in prof.items() if n}
        tot = sum(v["ms_per_step"] for v in kernels.values())
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        dominant = dict(name=dom, avg_launch_ms=kernels[dom]["avg_launch_ms"],
                        share_of_step=round(kernels[dom]["ms_per_step"] / tot, 4),
                        bound="fp64 vector ALU + LDS atomics (DESIGN.md section 5), not HBM"
                        if dom in ("k_scatter_sph", "k_gather_sph") else "hbm")

    sustained = None
    if args.sustained > 0 and world == 1:
        sustained = sustained_run(engine, q0, p0, q1, p1, eps, 1e3 * wall / args.steps, args.sustained, torch)
    throughput_mode = None
    if args.chains_per_gpu == 2 and world == 1:
        throughput_mode = two_chain_throughput(params, f, arrays, dev, local_rank, eps, args.steps, 1 if args.fp32 else 0)

    if rank == 0:
        N = params.N
        steps_total = args.steps * world
        value = steps_total / wall
        algo = ALGO_BYTES_PER_CELL_STEP // (2 if args.fp32 else 1)  # 272 N bytes per step with fp32 fields
        achieved = algo * N * args.steps / (gpu_ms * 1e-3) / 1e9  # per GPU, device time
        traffic, traffic_src = (pmc_traffic(params.Nx, "fp32" if args.fp32 else "fp64")
                                if (rsd and params.likelihood == 1 and headline_variant) else (None, None))
        out = {
            "metric": "HMC leapfrog steps/sec on %d^3 grid" % params.Nx + ("" if headline_variant else
                                                                        " (variant mk=%d calc_h=%d: not the headline)" % (args.mk, args.calc_h)),
            "value": round(value, 4),
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * wall / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.fp32 else "f64",
            "data": "synthetic",
            "config": {
                "workload": "%d^3 grid, L=200 Mpc/h, Gaussian prior, %s, likelihood=%d, mass kernel mk=%d (3 = SPH), calc_h=%d, mass_type=1, %s; one "
                            "trajectory of %d leapfrog steps per chain" % (params.Nx,
                                                                           ("Zel'dovich + plane-parallel RSD (reference "
                                                                            "behaviour of '2LPT+RSD', SURVEY M3)") if rsd else
                                                                           ("ALPT forward model (sfmodel=2, kth=4)"
                                                                            if args.alpt else "Zel'dovich"),
                                                                           params.likelihood, args.mk, args.calc_h,
                                                                           "fp32 field arrays" if args.fp32 else "fp64",
                                                                           args.steps),
                "grid": params.Nx, "chains": world, "parallelism": "independent chains, 1 per GPU",
                "rehearsal_single_device": bool(args.single_device),
                "eps": eps, "steps_done": int(done), "finite": finite,
            },
            "collective": collective,
            "roofline": {
                "bound": "hbm",
                "kernel": "whole leapfrog step (all kernels + 6 rocFFT transforms)",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_measured_in_run": False,  # PMC needs its own rocprofv3 passes (scripts/profile_round.sh)
                "algorithmic_bytes_per_step": algo * N,
                "device_ms_per_step": round(gpu_ms / args.steps, 4),
                "dominant_kernel": dominant,
                "kernels": kernels,
                # the two particle-mesh kernels are fp64-VALU bound, not HBM bound: reported against 78.6 TFLOP/s
                "valu": valu_roofline() if (rsd and params.likelihood == 1 and not args.fp32 and params.Nx == 256 and
                                            headline_variant) else None,
            },
        }
        if throughput_mode is not None:
            out["throughput_mode"] = throughput_mode
        if sustained is not None:
            out["sustained"] = sustained
        if world == 1 and not args.no_cpu_baseline:
            cpu_params = params
            cq0, cp0, carr = f["q0"], p0_host, arrays
            if args.cpu_nx and args.cpu_nx != params.Nx:
                from tests.util import Case  # small-grid sample of the same recipe
                c = Case(Nx=args.cpu_nx, L=200.0 * args.cpu_nx / params.Nx, likelihood=params.likelihood,
                         rsd_model=rsd)
                cpu_params, cq0, cp0, carr = c.p, c.q0, c.p0, c.arrays()
            out["cpu_baseline"], (q3o, p3o) = cpu_baseline(cpu_params, carr, cq0, cp0, eps)
            # parity at the benchmarked size: the oracle's 3-step trajectory (just timed) against a 3-step engine
            # trajectory on the same inputs, through the same entry point and kernel instantiations the bench times
            if cpu_params is params:
                e3q, e3p = torch.empty_like(q0), torch.empty_like(p0)
                engine.leapfrog_device(q0, p0, e3q, e3p, eps, 3)
                engine.sync()
                tol = 1e-4 if args.fp32 else 1e-11
                rq, rp = _rel_l2(e3q.cpu().numpy(), q3o), _rel_l2(e3p.cpu().numpy(), p3o)
                out["parity_at_size"] = dict(grid=params.Nx, steps=3, rel_l2_q=rq, rel_l2_p=rp, tolerance=tol,
                                             ok=bool(rq < tol and rp < tol),
                                             oracle="oracle/liboracle_omp.so (parity unpinned: no reference vectors exist)")
        print(json.dumps(out))
    sys.stdout.flush()
    if distributed:
        dist.barrier()
        if (collective.get("rccl_native") or {}).get("hung"):
            os._exit(0)  # a thread is stuck inside the probe's communicator set-up: do not wait for its teardown
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
