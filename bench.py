#!/usr/bin/env python3
"""bench.py -- minimizer iterations/s on BASELINE.json's genome-wide configuration.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher around it (WORLD_SIZE unset) the command starts its own N rank processes -- one per GPU, a
`python -m torch.distributed.run` child started BEFORE anything touches the GPU -- and passes their one JSON line through;
under a launcher (the driver's torchrun line) it is one of the ranks.  stdout carries the JSON line and nothing else
(what libraries print -- RCCL's version banner, Gloo's connection notes -- is sent to stderr).

One "step" is one accepted L-BFGS iteration (liblbfgs "progress" call) of the on-device minimizer
over a synthetic Hilbert-curve-initialised bead system.  N = 1: BASELINE config 3 (gw_200k: 200 000
beads, GW preset = EV + compartment blocks + container + lamina + bonds + angles + loops).  N > 1:
BASELINE config 4, one independent genome-wide replica per GPU with seeds 0..N-1 (the reference's
ensemble loop, run.py:471-485, is embarrassingly parallel): no data-path collective, "weak" scaling.
`value` is the whole-job rate: total iterations of all ranks / max-over-ranks wall time.  Every line also carries north_star's
STRONG-scaling curve as top-level keys `strong_1m_*` (gw_1m, ONE system on the N GPUs of the run: iterations/s, speed-up and
parallel efficiency against one GPU, ranks RCCL saw; at N = 1 one GPU minimizing gw_1m alone): the N = 1, 2, 4, 8 lines of a
scaling run form the curve by themselves.  At N > 1 the same JSON line
also carries a `dd` object: BASELINE config 5, ONE gw_1m system decomposed over the N GPUs (ghost-bead halo exchange +
one all-reduce per evaluation on RCCL), with its iterations/s, bytes exchanged per evaluation and parallel efficiency
against one GPU minimizing the same system (never part of `value`).

Extra objects on the JSON line (task contract): "roofline" for the dominant kernel (the cell-list
pair kernel) from HIP events recorded live on the library's stream inside the timed region, and
"cpu_baseline" = OpenMM's CPU platform when `import openmm` works on the box (probed at run time, kind "openmm"),
else (kind "port", with the import error recorded) the repository's own tuned CPU evaluation -- fp32, AVX-512 / AVX2, OpenMP,
the same L-BFGS (oracle/mmx_cpu_fast.c) -- with the plain fp64 restatement beside it, both timed on the host cores on a
bounded sample of the same workload.  With `--replicas-per-gpu 3` (N = 1) a third object, "replicas_per_gpu",
reports -- outside the timed region and never as part of `value` -- the aggregate rate of three independent replicas
sharing the GPU (`run_ensemble(..., concurrent=3)`): a single minimization leaves the GPU idle in its latency-bound
launches.  Off by default: its concurrent kernels would mix into a profile of the default command.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak (spec)
NB_BYTES_PER_BEAD = 32.0     # float4 position in + float4 force/energy out
FLOP_PER_PAIR = 30.0         # SURVEY.md 8(d) per-pair flop count for the VALU view


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="gw_200k")
    ap.add_argument("--n-beads", type=int, default=0, help="rescale the workload (parity/debug only)")
    ap.add_argument("--cutoff", type=float, default=0.6, help="pair cutoff in nm; <=0 = NoCutoff all-pairs")
    ap.add_argument("--jitter", type=float, default=0.0)
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="budget of the cpu_baseline leg; 0 disables")
    ap.add_argument("--profile-every", type=int, default=16,
                    help="HIP-event time every kernel slot of every k-th evaluation inside the timed region "
                         "(an event pair costs ~10 us of stream time: 16 keeps the perturbation < 1 %%)")
    ap.add_argument("--profile-nb-every", type=int, default=0,
                    help="HIP-event time the pair kernel alone in every k-th evaluation (one event pair); 0 = adaptive: "
                         "often enough for at least 8 samples in the timed region (2 for the driver's 20-step command)")
    ap.add_argument("--dd-timeout", type=float, default=240.0,
                    help="seconds the extra decomposed leg at N > 1 may take before it is abandoned: the line is printed "
                         "with the stage it was in and the process exits with code 3")
    ap.add_argument("--mode", choices=("ensemble", "dd"), default="ensemble",
                    help="what `value` is at N > 1: 'ensemble' = one gw_200k replica per GPU (config 4, weak scaling, no "
                         "collective; the default, with the gw_1m decomposed run reported beside it as the `dd` object); "
                         "'dd' = the timed headline itself is ONE system decomposed over the GPUs (config 5, RCCL)")
    ap.add_argument("--no-dd-leg", action="store_true", help="skip the gw_1m strong-scaling leg (N > 1: the decomposed run beside the ensemble; N = 1: one GPU minimizing gw_1m)")
    ap.add_argument("--replicas-per-gpu", type=int, default=0,
                    help="extra leg at N=1 (never part of `value`, off by default so that a profile of the default "
                         "command contains the timed minimization only): aggregate rate of this many replicas sharing "
                         "the GPU, e.g. 3")
    ap.add_argument("--serial-bonded", action="store_true",
                    help="bonded terms on the main stream instead of beside the cell build (A/B of the overlap)")
    ap.add_argument("--separate-bonded", action="store_true",
                    help="run backbone / loops / confinement as three kernels (per-kernel timing) instead of the "
                         "fused default")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="engine option set before the warm-up (A/B runs; repeatable), e.g. --option cell_slots=0")
    ap.add_argument("--nb-traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the pair kernel from a separate rocprofv3 --pmc run "
                         "(profiles/): copied into roofline.traffic")
    return ap.parse_args()


def replicas_per_gpu_leg(workload: str, n_beads, cutoff: float, device: int, k: int, iters: int) -> dict | None:
    """NOT part of `value`: aggregate iterations/s of k independent replicas (seeds 0..k-1) kept in flight on ONE GPU,
    one engine handle + stream + host thread each -- what `run_ensemble(..., concurrent=k)` uses for BASELINE config 4.
    Reported beside the headline because a single minimization leaves the GPU idle in its latency-bound launches."""
    import threading
    import time
    try:
        from multimm_amd import synthetic_system
        from multimm_amd.engine import engine_for
        systems = [synthetic_system(workload, seed=i, n_beads=n_beads or None, NB_CUTOFF=cutoff) for i in range(k)]
        engines = [engine_for(s, device=device) for s in systems]
        try:
            for e in engines:
                e.minimize(tolerance=0.0, max_iters=10)
            done = [0] * k
            gate = threading.Barrier(k + 1)

            def work(i):
                gate.wait()
                done[i] = engines[i].minimize(tolerance=0.0, max_iters=iters).iterations

            th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
            for t in th:
                t.start()
            gate.wait()
            t0 = time.perf_counter()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
        finally:
            for e in engines:
                e.close()
        return {"replicas": k, "iterations_each": iters, "value": sum(done) / dt, "unit": "iters/s (aggregate)",
                "note": "independent replicas sharing one GPU, after 10 warm-up iterations each; not the headline"}
    except Exception as exc:  # never let the extra leg break the bench line
        return {"error": repr(exc)}


def committed_traffic(args, n: int):
    """(bytes per launch, where from) of the pair kernel for this run's workload from profiles/nb_traffic.json, or (None, reason)."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "nb_traffic.json")))
    except Exception as exc:  # noqa: BLE001
        return None, f"profiles/nb_traffic.json unreadable: {exc!r}"
    ent = tab.get(args.workload)
    if ent is None:
        return None, f"no committed PMC pass for workload {args.workload} (have: {sorted(k for k in tab if not k.startswith('_'))})"
    if args.n_beads or args.cutoff != 0.6 or args.jitter != 0.0:
        return None, "no committed PMC pass for this size / cutoff / start (the committed ones: preset size, 0.6 nm, lattice start)"
    return ent["bytes_per_launch"], (f"committed rocprofv3 --pmc passes (profiles/nb_traffic.json: {ent['kernel']}, `{ent['command']}`; "
                                     f"2 x FETCH_SIZE + WRITE_SIZE)")


def cpu_baseline(system, budget_s: float) -> dict | None:
    """CPU baseline on this box's host cores, bounded sample.  North star: OpenMM's CPU platform -- probed at run time
    (oracle/openmm_probe.py): when ``import openmm`` works, the same System (built by this repo's host code) is
    minimized by LocalEnergyMinimizer with the same cutoff (kind "openmm"), plus NoCutoff as the reference runs it on a
    20 000-bead sample of the workload; otherwise (this image) the import error is recorded and the repository's own CPU
    code is timed (kind "port"): the tuned fp32 / SIMD / OpenMP evaluation with the same L-BFGS as `value`, and the plain
    fp64 restatement beside it (`restatement_fp64`)."""
    if budget_s <= 0:
        return None
    from oracle.openmm_probe import probe, openmm_minimize
    from oracle.oracle import Oracle, cpu_share, max_threads, set_threads
    hw_threads = max_threads()
    cores = min(hw_threads, cpu_share())      # what the job may use (affinity mask, cgroup quota), not what the box has
    set_threads(cores)
    mm, why = probe()
    if mm is not None:
        from multimm_amd import synthetic_system
        one = openmm_minimize(system, 1, "CPU", cores)           # sizes the sample
        iters = int(max(1, min(50, budget_s / max(one["seconds"], 1e-3))))
        res = openmm_minimize(system, iters, "CPU", cores)
        small = synthetic_system(system.name.split("@")[0], n_beads=20000, NB_CUTOFF=0.0)
        nocut = openmm_minimize(small, 3, "CPU", cores)
        return {"value": res["iters_per_s"], "unit": "iters/s", "cores": cores, "kind": "openmm",
                "sample": f"{iters} LocalEnergyMinimizer iterations ({res['seconds']:.1f} s) of the same {system.n_beads}-bead "
                          f"system on OpenMM's CPU platform, CutoffNonPeriodic at {system.ff.NB_CUTOFF} nm; {why}",
                "nocutoff_20k_iters_per_s": nocut["iters_per_s"]}
    orc = Oracle(system)
    # (a) the tuned port (oracle/mmx_cpu_fast.c): fp32, AVX-512 / AVX2 inner loop over cell-sorted neighbours, OpenMP over cells,
    # the same L-BFGS -- what a CPU platform does; this is `value`.  (b) the plain fp64 restatement (the checker of the parity
    # tests: scalar pow()/exp() per pair) for continuity with rounds 1-3.  Each gets half of the budget.
    from oracle.oracle import lib as orc_lib
    simd = {2: "AVX-512", 1: "AVX2+FMA", 0: "baseline x86-64"}.get(int(orc_lib().orc_fast_simd_level()), "?")
    orc.fast_eval()                      # (first call: thread pool, page faults)
    t0 = time.perf_counter()
    orc.fast_eval()
    t_eval = time.perf_counter() - t0
    iters = int(max(2, min(2000, 0.5 * budget_s / max(t_eval * 1.2, 1e-4))))
    t0 = time.perf_counter()
    _, st, swept = orc.fast_minimize(tolerance=0.0, max_iters=iters)
    dt = time.perf_counter() - t0
    out = {
        "value": st.iterations / dt, "unit": "iters/s", "cores": cores, "hardware_threads_of_the_box": hw_threads, "kind": "port",
        "implementation": f"tuned port: fp32, {simd} inner loop over cell-sorted neighbours (cells of cutoff / 2, 5x5x5 stencil, "
                          "every pair from both sides), OpenMP over cells, fp64 reductions, the same liblbfgs control flow; "
                          "checked against the fp64 restatement (tests/test_oracle.py::test_fast_cpu_baseline_*)",
        "sample": f"{st.iterations} L-BFGS iterations ({st.evaluations} evaluations, {dt:.1f} s) of the same "
                  f"{system.n_beads}-bead system from the same start, same cutoff",
        "evals_per_s": st.evaluations / dt,
        "lane_pairs_swept_per_s_per_core": swept / dt / max(cores, 1),
        "openmm": {"available": False, "probe": why,
                   "note": "the north star's >= 10 x OpenMM-CPU is NOT verified against OpenMM itself (not installable here); "
                           "`value` is this repository's own tuned CPU evaluation of the same force field"},
    }
    budget_left = 0.5 * budget_s
    t0 = time.perf_counter()
    orc.eval()
    t_eval = time.perf_counter() - t0
    iters = int(max(2, min(50, budget_left / max(t_eval * 1.3, 1e-3))))
    t0 = time.perf_counter()
    _, st = orc.minimize(tolerance=0.0, max_iters=iters)
    dt = time.perf_counter() - t0
    out["restatement_fp64"] = {
        "value": st.iterations / dt, "unit": "iters/s", "cores": cores,
        "implementation": "unvectorised fp64 full-shell restatement (oracle/mmx_oracle.c: pow()/exp() per pair, 27-cell sweep per "
                          "bead): the checker of the parity tests, a reference point of rounds 1-3, not a contender",
        "sample": f"{st.iterations} L-BFGS iterations ({st.evaluations} evaluations, {dt:.1f} s)",
        "evals_per_s": st.evaluations / dt,
    }
    return out


def dd_leg(args, rank: int, world: int, local_rank: int, tdev, barrier, progress: dict) -> dict:
    """BASELINE config 5 beside the headline (N > 1): ONE gw_1m system (1 000 000 beads), bead slices owned by the
    ranks, ghost-bead halo exchange + one fp64 all-reduce per evaluation on RCCL, issued by libmmx on its own stream.
    Returns the `dd` object of the JSON line: job iterations/s, the ranks RCCL saw, bytes exchanged per evaluation,
    and the parallel efficiency against ONE GPU minimizing the same system (timed on rank 0 right after).  Never part
    of `value`; any failure is reported inside the object instead of breaking the line."""
    import torch.distributed as dist
    from multimm_amd import synthetic_system
    from multimm_amd.engine import Engine, engine_for
    from multimm_amd.parallel import broadcast_bytes, reduce_job_stats
    out = {"workload": "gw_1m", "mode": "dd", "ranks": world}
    stage = "setup"

    def at(name):
        progress["stage"] = name
        return name
    try:
        if os.environ.get("MMX_BENCH_INJECT_DD_STALL"):  # tests / rehearsals of the watchdog: a leg that never comes back
            stage = at("injected stall")
            while True:
                time.sleep(1.0)
        system = synthetic_system("gw_1m", seed=0, NB_CUTOFF=args.cutoff)
        out["n_beads"] = system.n_beads
        eng = engine_for(system, device=local_rank, rank=rank, world=world)
        try:
            stage = at("rccl communicator")
            uid = broadcast_bytes(Engine.comm_unique_id() if rank == 0 else None, 128, device=tdev)
            eng.comm_init(uid)
            eng.set_option("profile", 16)      # HIP events around the kernel slots AND the collectives of every 16th evaluation
            stage = at("warm-up")
            if args.warmup > 0:
                eng.minimize(tolerance=0.0, max_iters=args.warmup)
            b0, x0 = eng.get_option("dd_bytes_sent"), eng.get_option("dd_exchanges")
            r0 = eng.get_option("dd_redecompositions")
            stage = at("timed minimization")
            barrier()
            t0 = time.perf_counter()
            st = eng.minimize(tolerance=0.0, max_iters=args.steps)
            barrier()
            dt = time.perf_counter() - t0
            dt, iters = reduce_job_stats(dt, st.iterations, "dd", device=tdev)
            nx = max(eng.get_option("dd_exchanges") - x0, 1.0)
            out.update({
                "value": iters / dt, "unit": "iters/s", "iterations": iters, "evaluations": st.evaluations,
                "ms_per_step": dt * 1e3 / max(iters, 1.0), "ranks_rccl_saw": world, "status": st.status,
                "halo_bytes_sent_per_evaluation_rank0": (eng.get_option("dd_bytes_sent") - b0) / nx,
                "allgather_bytes_it_replaces_per_rank": 16.0 * eng.n_own * (world - 1),
                "allreduce_bytes_per_evaluation": 8.0 * 59,
                "ghosts_rank0": eng.get_option("dd_ghosts"), "owned_rank0": eng.n_own,
                "list_rebuilds": eng.get_option("dd_redecompositions") - r0,
                "list_rebuilds_host_synchronous": eng.get_option("dd_sync_rebuilds"),
                "evaluations_voided_and_repeated": eng.get_option("dd_halts"),
                "rebuild_every": eng.get_option("dd_rebuild_every"),
                # spatial ownership: 62-bead segments re-assigned by recursive bisection while the structure deforms
                "segment_reassignments": eng.get_option("dd_reassignments"),
                "segment_reassignment_attempts": eng.get_option("dd_reassign_attempts"),
                "segments_moved": eng.get_option("dd_segments_moved"),
                # where an evaluation's time goes on rank 0 (HIP events on the library's stream, every 16th evaluation):
                # each collective from the end of the work before it to its own end = transfer + waiting for the peers
                "rank0_us_per_evaluation": {
                    "needmap_allgather": eng.get_option("dd_us_needmap_allgather"),
                    "halo_send_recv": eng.get_option("dd_us_halo_exchange"),
                    "allreduce_59_doubles": eng.get_option("dd_us_allreduce"),
                    "collective_samples": eng.get_option("dd_collective_samples"),
                    "kernel_slots": st.as_dict()["kernel_us_mean"],
                    "note": "cell_build contains the need-map all-gather and the halo exchange, reduce the all-reduce",
                },
            })
            # what one evaluation keeps rank 0's stream busy for: the HIP-event brackets of its kernel slots (pack + lists + halo
            # + cell build | pair kernel | tail | all-reduce + decision), collectives included -- the figure to hold against
            # scripts/dd_projection.py's compute-only critical path (profiles/)
            slots = [v for v in st.as_dict()["kernel_us_mean"].values() if v]
            out["critical_path_us"] = sum(slots) if slots else None
        finally:
            eng.close()
        stage = at("single-GPU reference")
        if rank == 0:
            with engine_for(system, device=local_rank) as e1:
                if args.warmup > 0:
                    e1.minimize(tolerance=0.0, max_iters=args.warmup)
                t0 = time.perf_counter()
                s1 = e1.minimize(tolerance=0.0, max_iters=args.steps)
                d1 = time.perf_counter() - t0
            out["single_gpu_iters_per_s"] = s1.iterations / d1
            out["parallel_efficiency"] = out["value"] / (world * out["single_gpu_iters_per_s"])
            out["speedup"] = out["value"] / out["single_gpu_iters_per_s"]
        if world > 1:
            dist.barrier()
    except Exception as exc:  # noqa: BLE001 -- the headline line must survive a failing extra leg
        out["error"] = f"{stage}: {exc!r}"
    return out


STRONG_KEYS = ("strong_1m_iters_per_s", "strong_1m_single_gpu_iters_per_s", "strong_1m_speedup", "strong_1m_parallel_efficiency",
               "strong_1m_ranks_rccl_saw", "strong_1m_critical_path_us")


def hoist_strong(out: dict, leg: dict | None, world: int) -> None:
    """north_star's strong-scaling curve as TOP-LEVEL keys of the line: gw_1m iterations/s of ONE system on the N GPUs of
    this run (N = 1: one GPU minimizing it alone), speed-up and parallel efficiency against one GPU, the ranks RCCL saw.
    The N = 1, 2, 4, 8 lines of a scaling run then form the curve by themselves.  None: not measured (the reason is in
    `dd` / `strong_1m_note`)."""
    leg = leg or {}
    out["strong_1m_iters_per_s"] = leg.get("value")
    out["strong_1m_single_gpu_iters_per_s"] = leg.get("single_gpu_iters_per_s")
    out["strong_1m_speedup"] = leg.get("speedup")
    out["strong_1m_parallel_efficiency"] = leg.get("parallel_efficiency")
    out["strong_1m_ranks_rccl_saw"] = leg.get("ranks_rccl_saw", world if leg.get("value") else None)
    out["strong_1m_critical_path_us"] = leg.get("critical_path_us")
    if leg.get("error"):
        out["strong_1m_note"] = leg["error"]


def guarded_leg(leg_fn, timeout_s: float, rank: int, world: int, out: dict, exit_fn=os._exit, emit_fn=None):
    """Runs the decomposed leg under a watchdog.  A leg that does not come back within `timeout_s` costs the `dd` object, not
    the line: rank 0 prints the line -- headline complete, the stage the leg was stuck in spelled out, the strong_1m_* keys
    None -- and then EVERY rank leaves with exit status 3: a rank hung in a collective holds its GPU, the launcher and the
    driver must see the job as failed (os._exit: a process stuck inside RCCL does not return)."""
    import threading
    progress = {"stage": "setup"}
    finished = threading.Event()
    emit_fn = emit_fn or emit

    def watchdog():
        if finished.wait(timeout_s):
            return
        if rank == 0:
            leg = {"workload": "gw_1m", "mode": "dd", "ranks": world,
                   "error": f"no result after {timeout_s:.0f} s (stage: {progress['stage']}); leg abandoned, exit status 3"}
            out["dd"] = leg
            hoist_strong(out, leg, world)
            emit_fn(out)
        else:
            time.sleep(3.0)
        exit_fn(3)
    threading.Thread(target=watchdog, daemon=True).start()
    leg = leg_fn(progress)
    finished.set()
    return leg


def strong_single_gpu(args, device: int) -> dict:
    """N = 1 point of the strong-scaling curve: gw_1m on this one GPU, same warm-up / steps as the headline, outside the timed
    region.  Skipped under rocprofv3 (its kernels carry the same names as the timed workload's and would mix into the kernel
    statistics of the profile)."""
    preload = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "") + os.environ.get("HSA_TOOLS_LIB", "")
    if "rocprof" in preload.lower():
        return {"error": "skipped under rocprofv3: the gw_1m kernels would mix into the kernel statistics of the timed workload"}
    try:
        from multimm_amd import synthetic_system
        from multimm_amd.engine import engine_for
        system = synthetic_system("gw_1m", seed=0, NB_CUTOFF=args.cutoff)
        with engine_for(system, device=device) as e1:
            if args.warmup > 0:
                e1.minimize(tolerance=0.0, max_iters=args.warmup)
            t0 = time.perf_counter()
            s1 = e1.minimize(tolerance=0.0, max_iters=args.steps)
            d1 = time.perf_counter() - t0
        v = s1.iterations / d1
        return {"value": v, "single_gpu_iters_per_s": v, "speedup": 1.0, "parallel_efficiency": 1.0, "ranks_rccl_saw": 1,
                "iterations": s1.iterations, "n_beads": system.n_beads}
    except Exception as exc:  # noqa: BLE001 -- never let the extra leg break the bench line
        return {"error": repr(exc)}


_JSON_FD = None


def emit(obj: dict) -> None:
    """The one JSON line, on the process's ORIGINAL stdout (fd 1 itself points at stderr by now, see main)."""
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def visible_gpu_count() -> int:
    """GPUs this process may use, WITHOUT initialising the HIP runtime (torch.cuda.device_count() falls through to
    hipGetDeviceCount on builds without amdsmi): the visibility variables first, else the KFD topology (nodes with SIMDs)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    n = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                for ln in f:
                    if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                        n += 1
    except OSError:
        return 0
    return n


def spawn_ranks(args) -> int:
    """`bench.py --gpus N` without a launcher: start N rank processes as ONE child (torch.distributed.run, which starts
    the ranks) before this process has made any GPU call -- it never makes one --, hand their JSON line through and
    return their exit code.  With fewer GPUs visible than ranks (a rehearsal on a one-GPU box) the ranks share devices:
    torch.distributed then runs on gloo, and the library's own RCCL communicator is told that every rank sits on a host
    of its own (NCCL_HOSTID, set per rank in main()): RCCL's duplicate-GPU check does not apply and the ranks talk through
    its socket transport over the loopback interface -- the real ncclAllGather / ncclSend / ncclRecv / ncclAllReduce call
    sites of the decomposed leg run with N ranks.  The line says so; its numbers are not a scaling measurement."""
    import socket
    import subprocess
    # A process whose GPU runtime is initialised must not start the launcher: under rocprofv3 the profiler's preloaded tool
    # library has done that before this line runs (the pool forbids such a hop after `--`).  Profile one rank directly.
    preload = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "") + os.environ.get("HSA_TOOLS_LIB", "")
    if "rocprof" in preload.lower():
        print("bench.py --gpus N starts its own rank processes and cannot do that from under rocprofv3; profile a rank "
              "directly: rocprofv3 ... -- python3 bench.py (one GPU), or give the launcher line to the profiler's child", file=sys.stderr)
        return 6
    ndev = visible_gpu_count()   # counted without touching HIP: this parent must never hold the GPU
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ndev < args.gpus:
        env.setdefault("MMX_DIST_BACKEND", "gloo")
        env["MMX_BENCH_REHEARSAL"] = f"{args.gpus} ranks share {ndev} GPU(s): launch-path rehearsal, not a scaling measurement"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return proc.returncode if line is not None or proc.returncode else 5


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # stdout is for the JSON line alone: whatever libraries write to fd 1 (RCCL's version banner, Gloo's connection
    # notes) goes to stderr from here on
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MMX_BENCH_REHEARSAL") and world > 1:
        # ranks share a GPU: each tells RCCL it is a host of its own, reachable through sockets on the loopback interface
        os.environ.update({"NCCL_HOSTID": f"mmx-rehearsal-rank-{rank}", "NCCL_IB_DISABLE": "1", "NCCL_SOCKET_IFNAME": "lo",
                           "NCCL_P2P_DISABLE": "1", "NCCL_SHM_DISABLE": "1", "NCCL_NET": "Socket"})
    import torch
    import torch.distributed as dist

    # MMX_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the launch path on a 1-GPU box)
    backend = os.environ.get("MMX_DIST_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    local_rank = local_rank % ndev
    tdev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    n_gpus = world

    from multimm_amd import synthetic_system
    from multimm_amd.engine import K_NONBONDED, K_CELL_BUILD, K_BACKBONE, K_LOOPS, K_CONFINE, K_LBFGS, K_REDUCE, engine_for

    dd = world > 1 and args.mode == "dd"
    system = synthetic_system(args.workload, seed=0 if dd else rank, n_beads=args.n_beads or None,
                              jitter=args.jitter, NB_CUTOFF=args.cutoff)
    if dd:
        from multimm_amd.engine import Engine
        from multimm_amd.parallel import broadcast_bytes
        eng = engine_for(system, device=local_rank, rank=rank, world=world)
        uid = broadcast_bytes(Engine.comm_unique_id() if rank == 0 else None, 128, device=tdev)
        eng.comm_init(uid)
    else:
        eng = engine_for(system, device=local_rank)
    eng.set_option("profile", 0)
    for kv in args.option:
        k, v = kv.split("=", 1)
        eng.set_option(k.strip(), float(v))
    eng.set_option("fused_bonded", 0 if args.separate_bonded else 1)
    if args.serial_bonded:
        eng.set_option("overlap_bonded", 0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: W untimed iterations (also pages in code objects and sizes the pair-kernel grid)
    if args.warmup > 0:
        eng.minimize(tolerance=0.0, max_iters=args.warmup)
    eng.set_option("profile", args.profile_every)
    # the pair kernel alone is sampled more often (one event pair): at least 8 samples whatever --steps is
    nb_every = args.profile_nb_every or max(1, min(args.profile_every, args.steps // 8))
    eng.set_option("profile_nb", nb_every)
    n3_before = eng.get_option("n3_launches")
    barrier()
    t0 = time.perf_counter()
    st = eng.minimize(tolerance=0.0, max_iters=args.steps)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0

    iters = st.iterations
    if world > 1:
        from multimm_amd.parallel import reduce_job_stats
        dt, total_iters = reduce_job_stats(dt, iters, args.mode, device=tdev)
    else:
        total_iters = float(iters)

    if rank == 0:
        d = st.as_dict()
        n = system.n_beads
        nb_us = d["kernel_us_mean"]["nonbonded"]
        census = None
        try:
            # (a census on a decomposed handle is a collective -- it rebuilds the ghost lists -- and only rank 0 is here)
            census = eng.nb_census() if args.cutoff > 0 and not dd else None
        except Exception:
            census = None
        roofline = None
        if nb_us:
            achieved = NB_BYTES_PER_BEAD * n / (nb_us * 1e-6) / 1e9
            traffic, traffic_note = args.nb_traffic_bytes, None
            if traffic is None:
                # per-launch HBM bytes of the same kernel on the same workload from the committed rocprofv3 --pmc passes (PMC
                # collection cannot share a run with the timed region): profiles/nb_traffic.json, one entry per workload
                # (scripts/r5_traffic.sh); anything else -- another size, cutoff or start -- has no committed pass: null + why
                traffic, traffic_note = committed_traffic(args, n)
            # the cell-list path picks its pair kernel per state (DESIGN_HISTORY.md 5c): name the one that ran, and how often
            n3_share = (eng.get_option("n3_launches") - n3_before) / max(st.kernel_launches[K_NONBONDED], 1)
            kname = ("k_nb_allpairs" if args.cutoff <= 0 else "k_nb_n3" if n3_share > 0.99 else "k_nb_clusters_j" if n3_share < 0.01
                     else f"k_nb_n3 ({100 * n3_share:.0f} % of the launches: the dense phase), then k_nb_clusters_j")
            # what bounds the kernel: VALU issue of the pair arithmetic + culls, and latency (4 waves per SIMD, a third of
            # the wave cycles waiting: DESIGN_HISTORY.md 5c) -- not HBM.  achieved/peak/frac are the HBM view the metric asks for
            # (algorithmic bytes / launch time against 8 TB/s); valu_view prices the same launch against the fp32 peak.
            roofline = {"bound": "valu+latency", "kernel": kname,
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": traffic_note if args.nb_traffic_bytes is None else "command line",
                        "traffic_over_algorithmic": (traffic / (NB_BYTES_PER_BEAD * n)) if traffic else None,
                        "launch_us": nb_us, "samples": int(st.kernel_samples[K_NONBONDED]),
                        "note": "achieved/peak/frac = HBM view (32 B/bead algorithmic bytes over the launch time); the kernel "
                                "itself is bound by VALU issue and latency, see valu_view and its wait counters"}
            if census:
                directed = census["pairs_within_cutoff"]     # the census walks every bead's full shell
                pairs = directed / 2.0                        # unique pairs: what the physics needs evaluated
                tf = pairs * FLOP_PER_PAIR / (nb_us * 1e-6) / 1e12
                roofline["valu_view"] = {"achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": tf / VALU_PEAK_TFLOPS, "unique_pairs_within_cutoff": pairs,
                                         "pair_candidates": census["pair_candidates"],
                                         "flop_per_pair": FLOP_PER_PAIR}
                # instruction counters of the half-shell kernel from the committed PMC passes: gw_200k at its lattice state only
                if args.workload == "gw_200k" and n == 200000 and args.cutoff == 0.6 and args.jitter == 0.0:
                    try:
                        roofline["valu_view"]["issue_from_committed_profile"] = json.load(
                            open(os.path.join(ROOT, "profiles", "nb_valu.json")))
                        roofline["valu_view"]["instruction_budget"] = "profiles/r05/n3_instruction_budget.txt"
                    except Exception as exc:  # noqa: BLE001
                        roofline["valu_view"]["issue_from_committed_profile"] = None
                        roofline["valu_view"]["issue_note"] = repr(exc)
                else:
                    roofline["valu_view"]["issue_from_committed_profile"] = None
                    roofline["valu_view"]["issue_note"] = "no committed PMC pass for this workload / state (gw_200k, 0.6 nm, lattice start only)"
        kern = {k: v for k, v in d["kernel_us_mean"].items() if v}
        # default: backbone + loops + confinement are ONE pass (38 B/bead + 64 B/loop) that rides in the launch of the
        # cell scan, i.e. inside the "cell_build" slot, which also forms the L-BFGS direction (168 B/bead) in its pack;
        # --serial-bonded books the pass under "confine", --separate-bonded times the three kernels individually.
        # "lbfgs" = k_history: 228 B/bead.
        fused_bytes = 38.0 * n + 64.0 * system.n_loops
        in_scan = not (args.serial_bonded or args.separate_bonded) and args.cutoff > 0
        alg_bytes = {"cell_build": (56.0 + 168.0) * n + (fused_bytes if in_scan else 0.0), "backbone": 25.0 * n,
                     "loops": 64.0 * system.n_loops, "confine": 25.0 * n if args.separate_bonded else fused_bytes,
                     "lbfgs": 228.0 * n}
        kernel_gbs = {k: alg_bytes[k] / (kern[k] * 1e-6) / 1e9 for k in alg_bytes if k in kern}
        ms_per_step = dt * 1e3 / max(iters, 1)
        out = {
            "metric": "minimizer iters/sec @ genome-wide N beads, 1/2/4/8 GPUs; achieved HBM GB/s",
            "value": total_iters / dt,
            "unit": "iters/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if dd else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{system.name}: {n} beads, Hilbert-curve start, "
                            f"{'GW preset (EV+COB+container+lamina+bonds+angles+loops)' if 'gw' in args.workload else 'EV+bonds+angles+loops'}, "
                            f"{system.n_loops} loops, pair cutoff {args.cutoff} nm"
                            + ("; ONE system, bead slices owned by the GPUs, ghost-bead halo exchange (ncclSend/ncclRecv of the "
                               "beads the peers' need-maps ask for) + one fp64 all-reduce per evaluation on RCCL"
                               if dd else "; one independent replica per GPU (seeds 0..N-1), no collective"
                               if world > 1 else ""),
                "n_beads": n, "cutoff_nm": args.cutoff, "replicas": 1 if dd else world, "mode": args.mode if world > 1 else "single",
            },
            "iterations": iters, "evaluations": st.evaluations,
            "evals_per_s": st.evaluations * (1 if dd else world) / dt,
            "status": st.status, "e_initial": st.e_initial, "e_final": st.e_final, "rms_force": st.rms_force,
            "kernel_us_mean": kern,
            "bonded_kernels": ("separate" if args.separate_bonded else "one pass, confine slot" if not in_scan
                               else "one pass inside the cell build's launch (k_build_direct; cell_build slot)"),
            "kernel_algorithmic_GBps": kernel_gbs,
            "roofline": roofline,
        }
        if os.environ.get("MMX_BENCH_REHEARSAL"):
            out["rehearsal"] = os.environ["MMX_BENCH_REHEARSAL"]
        out["cpu_baseline"] = cpu_baseline(system, args.cpu_seconds) if n_gpus == 1 else None
        if out["cpu_baseline"] and out["cpu_baseline"].get("value"):
            # (a reported ratio, not the target's proof: the CPU side is this repository's own code unless kind == "openmm")
            out["cpu_baseline"]["gpu_over_cpu_iters_per_s"] = out["value"] / out["cpu_baseline"]["value"]
    eng.close()
    if rank == 0 and n_gpus == 1 and args.cpu_seconds > 0 and args.cutoff > 0:
        # what the cutoff changes against the reference's NoCutoff semantics, measured in this run on chr1_50k (BASELINE
        # config 2) with the exact all-pairs kernel as the yardstick -- outside the timed region, like the CPU baseline
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            from cutoff_tolerance import compare
            r = compare("chr1_50k", None, args.cutoff)
            out["truncation"] = {
                "workload": "chr1_50k", "cutoff_nm": args.cutoff, "against": "NoCutoff (exact all-pairs kernel), same start",
                "start": {k: r["start"][k] for k in ("dE_total", "dE_per_bead", "e_total_nocutoff", "dF_rms", "dF_max",
                                                      "dF_rel_l2", "F_rms_nocutoff")},
                "converged": {k: r["converged"][k] for k in ("iterations_cutoff", "iterations_nocutoff", "dE_rel", "d_rg_rel",
                                                              "d_bond_mean_nm")},
                "asserted_by": "tests/test_gpu_cutoff.py",
            }
        except Exception as exc:  # noqa: BLE001
            out["truncation"] = {"error": repr(exc)}
    if world > 1 and not dd and not args.no_dd_leg:  # every rank takes part; rank 0 carries the result
        # The RCCL path of the decomposed run has never met more than one rank on hardware: a watchdog makes sure
        # that a leg which does not come back costs the `dd` object, not the line -- and ends the job with exit status 3.
        leg = guarded_leg(lambda progress: dd_leg(args, rank, world, local_rank, tdev, barrier, progress), args.dd_timeout,
                          rank, world, out if rank == 0 else {})
        if rank == 0:
            out["dd"] = leg
            hoist_strong(out, leg, world)
    elif rank == 0 and dd:  # --mode dd: the headline itself is the decomposed run of whatever workload was asked for
        hoist_strong(out, {"value": out["value"], "ranks_rccl_saw": world} if args.workload == "gw_1m" else None, world)
    elif rank == 0 and n_gpus == 1 and not args.no_dd_leg:
        hoist_strong(out, strong_single_gpu(args, local_rank), 1)
    elif rank == 0:
        hoist_strong(out, None, world)
    if rank == 0:
        if n_gpus == 1 and args.replicas_per_gpu > 1:
            out["replicas_per_gpu"] = replicas_per_gpu_leg(args.workload, args.n_beads, args.cutoff, local_rank,
                                                           args.replicas_per_gpu, args.steps)
        emit(out)
    if world > 1:
        import threading
        threading.Timer(60.0, lambda: os._exit(4)).start()  # a rank that lost its peers must not keep the launcher waiting
        dist.barrier()
        dist.destroy_process_group()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
