#!/usr/bin/env python3
"""bench.py -- Mrays/s of the render launch on the synthetic 1M-triangle scene at 1920x1080.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a launcher: it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child before anything touches the GPU
(it never imports torch) and exits with the child's code.  Under torch.distributed.run (the driver's own
N > 1 launch) the process is a worker and WORLD_SIZE must equal --gpus.
One "step" = one hrt_render_launch of the whole frame at the configured spp (BASELINE config C4:
1M random triangles, 1920x1080, 256 spp), scene/BVH/RNG states already resident in HBM.  For N > 1
the frame is split into interleaved 8-row stripes (one tile per GPU, BVH replicated) and each step
ends with the RCCL reduce of the per-tile radiance into rank 0's frame.

Prints ONE JSON line on rank 0 with the metric, a `roofline` object for the dominant kernel
(closest-hit traversal; algorithmic bytes per ray x rays / HIP-event time) and a `cpu_baseline`
object (the CPU oracle, OpenMP, on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
RAY_BYTES, HIT_BYTES, NODE_BYTES, PRIM_BYTES = 32, 20, 80, 48      # DESIGN.md "algorithmic bytes"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C4", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--spp", type=int, default=0, help="override the config's samples per pixel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-builder", action="store_true", help="skip the short run on the other builder's tree (config.alt_builder)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def launcher_argv(gpus, env, argv, port=None):
    """The launcher decision, as a pure function (tests/test_bench_cpu.py): None when this process must do the work itself
    (one GPU, or already a worker of torch.distributed.run), else the command line of the N-rank child."""
    if gpus <= 1 or "WORLD_SIZE" in env or "RANK" in env or "LOCAL_RANK" in env:
        return None
    if port is None:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]


def check_world(gpus, world):
    """A worker whose WORLD_SIZE disagrees with --gpus would print a line labelled with the wrong GPU count."""
    if world != gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but WORLD_SIZE={world}: pass the same number to both (a bare "
                         f"`python bench.py --gpus N` starts its N ranks by itself)")


def build_id():
    """What make wrote beside libhrt.so: a hash of the kernel sources the library was built from."""
    try:
        return (ROOT / "nvidia-optix-ray-tracer_amd" / "lib" / "BUILD_ID").read_text().strip()
    except OSError:
        return None


def profile_fields(profiles_dir, running):
    """(traffic, traffic_source, issue) of the roofline block from the newest committed PMC summaries (tools/profile_final.sh).  A
    figure is quoted only when its file was taken on the build that is running (`running` = lib/BUILD_ID); otherwise the figure
    is None and the source is marked stale -- the bench never prints the counters of another build as its own."""
    def newest(pattern):
        for f in sorted(Path(profiles_dir).glob(pattern), reverse=True)[:1]:
            try:
                d = json.loads(f.read_text())
            except Exception:
                return None, None, None
            return d, f"profiles/{f.name}", d.get("provenance", {}).get("build")
        return None, None, None
    traffic = traffic_source = issue = None
    d, name, b = newest("r*_traverse_traffic.json")
    if d is not None:
        stale = running is None or b != running
        traffic = None if stale else d.get("fabric_bytes_per_launch")
        traffic_source = {"file": name, "build": b, "running_build": running, "stale": stale, "tcc_hit_rate": None if stale else d.get("tcc_hit_rate"),
                          "what": "bytes the L2s requested from the fabric per launch (read + write); Infinity-Cache hits included: an upper bound on HBM bytes"}
    d, name, b = newest("r*_pmc_fused_kernel.json")
    if d is not None and "valu" in d:
        stale = running is None or b != running
        issue = {"file": name, "build": b, "running_build": running, "stale": stale}
        c = d.get("counters_mean_per_launch", {})
        cyc = d["valu"].get("kernel_cycles")
        n_inst = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"))
        if not stale and cyc and n_inst:
            cpi = cyc * 1024.0 / n_inst
            issue.update({
                "instructions_per_launch": n_inst, "simd_cycles_per_instruction": round(cpi, 3),
                "floor_cycles_per_instruction": ISSUE_FLOOR_CYCLES,
                "issue_frac": round(ISSUE_FLOOR_CYCLES / cpi, 4),
                "lanes_active_frac": round(d["valu"]["lanes_active_frac"], 4),
                "useful_lane_frac": round(ISSUE_FLOOR_CYCLES / cpi * d["valu"]["lanes_active_frac"], 4),
                "wave_time_split": {k: round(v, 4) for k, v in (d["valu"].get("wave_time_split") or {}).items()},
                "what": "issue_frac = 2.4 SIMD cycles per instruction (measured floor, any kind) / (kernel cycles x 1024 SIMDs / instructions); "
                        "lanes_active = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); useful_lane_frac = their product"})
    return traffic, traffic_source, issue


ISSUE_FLOOR_CYCLES = 2.4        # SIMD cycles per instruction with >= 2 waves issuing side by side (profiles/r02_valu_pipes_microbench.txt)


def effective_cpus():
    """How many CPUs this process may really use, and why: the OpenMP default is the machine's logical CPU count, which on a shared
    box is far more than the affinity mask / the cgroup's quota gives (round 3: 128 threads on the GPU box ran 0.025 Mrays/s each
    against 0.085 in the 8-CPU build container).  Returns (n, detail)."""
    detail = {"os_cpu_count": os.cpu_count()}
    n = os.cpu_count() or 1
    try:
        detail["sched_affinity"] = len(os.sched_getaffinity(0))
        n = min(n, detail["sched_affinity"])
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                detail["cgroup_cpu_max"] = " ".join(txt)
                if txt[0] != "max":
                    n = min(n, max(1, int(np.ceil(int(txt[0]) / int(txt[1])))))
            else:
                quota = int(txt[0])
                period = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                detail["cgroup_cfs_quota_us"], detail["cgroup_cfs_period_us"] = quota, period
                if quota > 0:
                    n = min(n, max(1, int(np.ceil(quota / period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n), detail


def cpu_baseline(hrt, scene, target_seconds, renderer=None):
    """The CPU oracle (kind "port": the reference has no CPU path to build) on a bounded sample of
    the same workload: the same frame at a reduced number of samples per pixel (or a subset of its
    rows when even 1 spp would take too long), sized for ~target_seconds, on as many threads as the process
    really has CPUs (effective_cpus).  With a renderer the oracle walks THE PRODUCT'S TREE -- the packed BVH8 the
    GPU has just been timed on, fetched with hrt_tlas_download ("same BVH bytes as the GPU
    run"; the closest hit is canonical, so the image is the oracle's own) -- and the GPU renders the very same
    sample afterwards; the two images are compared (the metric's "image L-inf vs ref"): returns (cpu_baseline, parity)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_py
    W, H = scene["width"], scene["height"]
    omp_default = oracle_py.lib().oracle_num_threads()
    threads, cpu_detail = effective_cpus()
    threads = min(threads, omp_default) if os.environ.get("OMP_NUM_THREADS") else threads
    oracle_py.lib().oracle_set_threads(threads)
    osc = oracle_py.OracleScene(scene)
    tree = "oracle/oracle.c's own median-split BVH2"
    if renderer is not None and os.environ.get("HRT_BENCH_CPU_OWN_TREE") != "1":
        import ctypes as C
        blob = hrt.BvhBlob()
        if renderer.lib.hrt_tlas_download(renderer.ctx, renderer.tlas, C.byref(blob)) == 0:
            nodes = np.ctypeslib.as_array(C.cast(blob.nodes, C.POINTER(C.c_uint8)), shape=(blob.n_nodes * 80,)).copy()
            prims = np.ctypeslib.as_array(C.cast(blob.triangles, C.POINTER(C.c_uint8)), shape=(max(blob.n_triangles, 1) * 48,)).copy()
            tree = f"the product's packed BVH8 as timed on the GPU (hrt_tlas_download: {blob.n_nodes} nodes, {blob.n_triangles} records, {len(nodes) + len(prims)} bytes)"
            renderer.lib.hrt_host_free(C.byref(blob))
            osc.attach_bvh8(nodes, prims)
    states = oracle_py.rng_init(W, H, hrt.scenes.SEED_SALT)
    probe_rows = np.arange(4, H, 16, dtype=np.uint32)                    # calibrate on 1/16 of the rows
    t0 = time.perf_counter()
    r = osc.render(W, H, states, 1, rows=probe_rows, want_linear=False)
    dt = max(time.perf_counter() - t0, 1e-6)
    frame_seconds = dt * H / len(probe_rows)                             # estimated time of one full 1-spp frame
    if frame_seconds > target_seconds:
        step = int(np.ceil(frame_seconds / target_seconds))
        rows, spp = np.arange(0, H, step, dtype=np.uint32), 1
    else:
        rows, spp = None, int(np.clip(target_seconds / frame_seconds, 1, scene["spp"]))
    states = oracle_py.rng_init(W, H, hrt.scenes.SEED_SALT)                # the timed sample starts from the seeded streams
    t0 = time.perf_counter()
    r = osc.render(W, H, states, spp, rows=rows, want_linear=renderer is not None)
    dt = max(time.perf_counter() - t0, 1e-6)
    osc.close()
    n_rows = H if rows is None else len(rows)
    base = {"value": round(r["rays"] / dt / 1e6, 4), "unit": "Mrays/s", "cores": int(threads), "kind": "port",
            "sample": f"{n_rows} of {H} rows x {W} px, {spp} spp, {r['rays']} rays in {dt:.1f} s, oracle/oracle.c OpenMP x{threads} walking {tree}",
            "per_thread": round(r["rays"] / dt / 1e6 / threads, 4), "node_visits_per_ray": round(r["node_visits"] / max(r["rays"], 1), 2),
            "cpus": dict(cpu_detail, omp_default_threads=int(omp_default), threads_used=int(threads))}
    if renderer is None:
        return base, None
    # the same sample on the GPU, from the same seeded streams
    renderer.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False, linear=True)
    renderer.reset_stats()
    tile = None if rows is None else hrt.Tile(0, H, 1, int(rows[1] - rows[0]) if len(rows) > 1 else H, 0)
    renderer.render(spp, tile=tile)
    sel = slice(None) if rows is None else rows
    g_lin, g_col = renderer.linear.cpu().numpy()[sel], renderer.color.cpu().numpy()[sel]
    parity = {"against": "oracle/oracle.c on the cpu_baseline sample", "pixels": int(g_lin.shape[0] * g_lin.shape[1]), "spp": int(spp),
              "linear_radiance_bit_exact": bool(np.array_equal(g_lin.view(np.uint32), r["linear"][sel].view(np.uint32))),
              "linf_color": float(np.abs(g_col - r["color"][sel]).max()),
              "rays_equal": bool(renderer.stats().rays == r["rays"])}
    return base, parity


def main():
    args = parse()
    child = launcher_argv(args.gpus, os.environ, sys.argv[1:])
    if child is not None:                       # launcher: nothing in this process has touched (or will touch) the GPU
        import subprocess
        rc = subprocess.call(child)
        if rc != 0:
            print(f"bench.py: the {args.gpus}-rank run failed with exit code {rc}", file=sys.stderr)
        sys.exit(rc)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    check_world(args.gpus, world)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs for a box with fewer GPUs than ranks (never set by the driver): HRT_BENCH_BACKEND=gloo lets several
    # ranks share a card (RCCL refuses duplicate devices), HRT_BENCH_SHARE_GPU=1 maps local ranks onto the visible cards
    backend = os.environ.get("HRT_BENCH_BACKEND", "nccl")
    if os.environ.get("HRT_BENCH_SHARE_GPU") == "1":
        local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def barrier():
        if backend == "nccl":
            dist.barrier(device_ids=[local_rank])
        else:
            dist.barrier()

    hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
    scene = hrt.scenes.BASELINE_CONFIGS[args.config]()
    if args.spp > 0:
        scene["spp"] = args.spp
    W, H, spp = scene["width"], scene["height"], scene["spp"]

    # The scene is static, so trace speed is preferred to build speed, as the reference does for its geometry
    # (OPTIX_BUILD_FLAG_PREFER_FAST_TRACE, RendererImpl.cu:94): HRT_CTX_FAST_TRACE = a tree with spatial splits, built on the device (top-down
    # SAH splits of references, then PLOC within the cells: build_split.hip; HRT_FAST_TRACE_BUILD=host: the host's binned-SAH builder, the same
    # rules, ~1.3 s).  HRT_BENCH_DEVICE_BUILD=1 times the frame on the default tree instead (the same top-down phase without spatial splits,
    # refittable: 8 ms, ~10 % more node visits per ray); the line carries the other builders' rates as config.alt_builders.
    device_build = os.environ.get("HRT_BENCH_DEVICE_BUILD") == "1"
    split_on_device = os.environ.get("HRT_FAST_TRACE_BUILD", "device") == "device"
    LABEL_PLOC = "device top-down SAH (object splits) + PLOC in the cells: the default, refittable build (build_split.hip + build.hip)"
    LABEL_PLOC_ALONE = "device PLOC alone (build.hip, HRT_BUILD_TOPDOWN=0)"
    LABEL_SPLIT = {True: "device top-down SAH with spatial splits + PLOC in the cells (build_split.hip, HRT_CTX_FAST_TRACE)",
                   False: "host binned SAH with spatial splits (bvh8_build.cpp, HRT_CTX_FAST_TRACE, HRT_FAST_TRACE_BUILD=host)"}
    # HRT_BENCH_NO_TIMING=1: no per-kernel HIP events (wavefront mode then replays its samples from a hipGraph; the roofline block has no kernel time)
    timing = 0 if os.environ.get("HRT_BENCH_NO_TIMING") == "1" else hrt.CTX_TIMING
    r = hrt.Renderer(local_rank, timing | (0 if device_build else hrt.CTX_FAST_TRACE))
    t0 = time.perf_counter()
    r.load_scene(scene)
    build_s = time.perf_counter() - t0
    r.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False)
    tile = hrt.tile_for_rank(H, rank, world) if world > 1 else None

    def step():
        if world > 1:
            r.color.zero_()                                       # rows of the other ranks must be zero for the sum
        r.render(spp, tile=tile, sync=False)
        if world > 1:
            hrt.reduce_tiles(r.color, dst=0)                      # per-tile radiance -> rank 0 (x + 0 is exact)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    r.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    own_elapsed = time.perf_counter() - t0          # this rank's own work done (before it waits for the others at the barrier)
    fence()
    elapsed = time.perf_counter() - t0
    st = r.stats()

    # max over ranks of the time, sum over ranks of the rays; and every rank's own time up to the end of ITS work (N > 1: the tail
    # imbalance of the tile split -- a rank whose stripes hold the long sample chains -- shows as min against max)
    own = [0.0] * world
    own[rank] = own_elapsed
    per_rank = torch.tensor(own, dtype=torch.float64, device=dev)
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rays = torch.tensor([float(st.rays), float(st.rays_closest), st.kernel_ms[hrt.K_TRAVERSE],
                         float(st.kernel_launches[hrt.K_TRAVERSE]), 1.0], dtype=torch.float64, device=dev)   # [4]: one per rank
    if world > 1:
        if backend != "nccl":
            tt, rays = tt.cpu(), rays.cpu()
            per_rank = per_rank.cpu()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    rank_elapsed = [float(x) for x in per_rank.tolist()]
    elapsed = float(tt.item())
    total_rays = float(rays[0].item())
    ranks_seen = int(round(float(rays[4].item())))
    if ranks_seen != args.gpus:
        raise SystemExit(f"bench.py: {ranks_seen} ranks took part in the all_reduce, --gpus says {args.gpus}")

    # ---- per-ray node / primitive counts (canonical walk order): one extra untimed pass ----
    r.set_flags(hrt.CTX_COUNT)
    r.reset_stats()
    r.render(2, tile=tile, sync=True)
    sc = r.stats()
    nodes_per_ray = sc.node_visits_closest / max(sc.rays_closest, 1)
    prims_per_ray = sc.prim_tests_closest / max(sc.rays_closest, 1)
    bytes_per_ray = RAY_BYTES + NODE_BYTES * nodes_per_ray + PRIM_BYTES * prims_per_ray + HIT_BYTES
    any_nodes = (sc.node_visits - sc.node_visits_closest) / max(sc.rays_any, 1)
    any_prims = (sc.prim_tests - sc.prim_tests_closest) / max(sc.rays_any, 1)
    bytes_per_any_ray = RAY_BYTES + NODE_BYTES * any_nodes + PRIM_BYTES * any_prims + HIT_BYTES

    if rank == 0:
        fused = st.kernel_launches[hrt.K_PATHS] > 0
        if fused:
            # fused path mode: ONE launch per step renders every sample of every pixel; rays live in registers, so
            # the algorithmic traffic is the BVH bytes a ray has to read (no ray / hit records)
            trav_ms = st.kernel_ms[hrt.K_PATHS]
            trav_launches = max(int(st.kernel_launches[hrt.K_PATHS]), 1)
            b_closest = NODE_BYTES * nodes_per_ray + PRIM_BYTES * prims_per_ray
            b_any = NODE_BYTES * any_nodes + PRIM_BYTES * any_prims
            which = os.environ.get("HRT_FUSED", "1")
            kernel_name = {"2": "k_traverse<FUSED> (round 1's persistent path kernel)"}.get(
                which, "k_fused (persistent path kernel: generate + traverse + shade + accumulate)")
        else:
            # wavefront mode: every traverse launch of the timed region (closest-hit rays and the depth-5 any-hit rays)
            trav_ms = st.kernel_ms[hrt.K_TRAVERSE] + st.kernel_ms[hrt.K_TRAVERSE_ANY]
            trav_launches = max(int(st.kernel_launches[hrt.K_TRAVERSE] + st.kernel_launches[hrt.K_TRAVERSE_ANY]), 1)
            b_closest, b_any = bytes_per_ray, bytes_per_any_ray
            kernel_name = "k_traverse"
        avg_launch_ms = trav_ms / trav_launches
        rays_per_launch = st.rays / trav_launches
        bytes_per_launch = (b_closest * st.rays_closest + b_any * st.rays_any) / trav_launches
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        # memory-side traffic and the instruction-issue view come from the PMC passes committed for THIS workload (tools/profile_final.sh:
        # separate --pmc runs, which cannot share a process with the timed region).  They are only quoted when the file was taken on
        # the build that is running (lib/BUILD_ID = hash of the kernel sources); otherwise they are null and marked stale.
        traffic = traffic_source = issue = None
        if args.config == "C4" and world == 1 and fused and args.spp == 0:
            traffic, traffic_source, issue = profile_fields(ROOT / "profiles", build_id())
        out = {
            "metric": "Mrays/s at 1920x1080, 1M-tri scene",
            "value": round(total_rays / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
            "rank_ms_per_step": {"min": round(min(rank_elapsed) / max(args.steps, 1) * 1e3, 3), "max": round(max(rank_elapsed) / max(args.steps, 1) * 1e3, 3),
                                 "per_rank": [round(x / max(args.steps, 1) * 1e3, 3) for x in rank_elapsed]},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{scene['name']}: {sum(len(i.get('vertices', ())) for i in scene['instances'])} triangles, "
                                   f"{W}x{H}, {spp} spp, depth {5}, tile-split x{world} (8-row stripes, BVH replicated)",
                       "rays_per_step": int(total_rays / max(args.steps, 1)), "rays_per_path": round(st.rays / max(st.paths, 1), 4),
                       "bvh_nodes": int(st.bvh_nodes), "bvh_bytes": int(st.bvh_bytes), "bvh_build_s": round(build_s, 3),
                       "bvh_builder": LABEL_PLOC if device_build else LABEL_SPLIT[split_on_device]},
            # What bounds the kernel is INSTRUCTION ISSUE (`bound`, `issue_frac`, `useful_lane_frac`: DESIGN.md section 4.1), not memory:
            # the tree of this scene is resident in L2 / Infinity Cache.  `achieved` / `peak` / `frac` stay the contract's figure
            # (ALGORITHMIC bytes -- 80 B per node visit + 48 B per primitive test -- over the launch time over the HBM
            # peak; repeated as `algorithmic_frac`), `traffic` / `measured_frac` are what reached the fabric.
            "roofline": {"bound": "instruction-issue", "kernel": kernel_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "algorithmic_frac": round(achieved / HBM_PEAK_GBS, 4),
                         "issue_frac": issue.get("issue_frac") if issue else None,
                         "useful_lane_frac": issue.get("useful_lane_frac") if issue else None,
                         "issue": issue,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "measured_frac": round(traffic / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and avg_launch_ms > 0 else None,
                         "note": "frac is algorithmic bytes over the HBM peak, served mostly by L2 / Infinity Cache (tree resident); the kernel is issue-bound",
                         "bytes_per_ray": round(b_closest, 1), "nodes_per_ray": round(nodes_per_ray, 3),
                         "prims_per_ray": round(prims_per_ray, 3), "bytes_per_any_hit_ray": round(b_any, 1),
                         "bytes_per_launch": round(bytes_per_launch, 0), "avg_launch_ms": round(avg_launch_ms, 4),
                         "rays_per_launch": round(rays_per_launch, 1), "launches": trav_launches},
            "kernel_ms": {hrt.KERNEL_NAMES[k]: round(st.kernel_ms[k], 3) for k in range(hrt.K_COUNT)},
        }
        if world == 1 and not args.no_alt_builder:
            # the same frame on the trees of the OTHER builders (a context each; a few steps are enough for a rate)
            # (label, context flag, environment for the context's creation)
            alts = [(LABEL_PLOC, 0, {})] if not device_build else [(LABEL_SPLIT[split_on_device], hrt.CTX_FAST_TRACE, {})]
            alts.append((LABEL_SPLIT[not split_on_device], hrt.CTX_FAST_TRACE, {"HRT_FAST_TRACE_BUILD": "host" if split_on_device else "device"}))
            alts.append((LABEL_PLOC_ALONE, 0, {"HRT_BUILD_TOPDOWN": "0"}))
            out["config"]["alt_builders"] = []
            for label, flag, env in alts:
                saved = {k: os.environ.get(k) for k in env}
                os.environ.update(env)                                   # (read when the context is created)
                try:
                    alt = hrt.Renderer(local_rank, hrt.CTX_TIMING | flag)
                finally:
                    for k, v in saved.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                t0 = time.perf_counter()
                alt.load_scene(scene)
                alt_build_s = time.perf_counter() - t0
                alt.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False)
                alt.render(spp, sync=True)
                alt.reset_stats()
                alt_steps = max(1, min(args.steps, 3))
                t0 = time.perf_counter()
                for _ in range(alt_steps):
                    alt.render(spp, sync=False)
                torch.cuda.synchronize(dev)
                alt_dt = time.perf_counter() - t0
                sa = alt.stats()
                out["config"]["alt_builders"].append({"bvh_builder": label, "value": round(sa.rays / alt_dt / 1e6, 3), "unit": "Mrays/s", "steps": alt_steps,
                                                      "bvh_nodes": int(sa.bvh_nodes), "bvh_build_s": round(alt_build_s, 3)})
                alt.close()
        if world == 1 and not args.no_alt_builder:
            # the same frame with HRT_CTX_REUSE_PRIMARY (include/hrt.h): a pixel's primary ray is traversed once per launch instead of
            # once per sample -- the same image bit for bit in less time.  Its rays are the TRAVERSED ones only (a third fewer), so
            # its Mrays/s is not this line's `value` and is not printed: the gain is the step time.
            r.set_flags(0)
            r.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False)          # (the same RNG states for both)
            r.render(min(spp, 8), tile=tile, sync=True)
            without = r.color.clone()
            r.set_flags(hrt.CTX_REUSE_PRIMARY)
            r.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False)
            r.render(min(spp, 8), tile=tile, sync=True)
            same = bool(torch.equal(without.view(torch.int32), r.color.view(torch.int32)))
            r.reset_stats()
            reuse_steps = max(1, min(args.steps, 3))
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reuse_steps):
                r.render(spp, tile=tile, sync=False)
            torch.cuda.synchronize(dev)
            reuse_dt = time.perf_counter() - t0
            sr = r.stats()
            out["config"]["primary_reuse"] = {"flag": "HRT_CTX_REUSE_PRIMARY (off for `value`)", "steps": reuse_steps, "ms_per_step": round(reuse_dt / reuse_steps * 1e3, 3),
                                              "ms_per_step_without": out["ms_per_step"], "traversed_rays_per_step": int(sr.rays // reuse_steps),
                                              "paths_per_step": int(sr.paths // reuse_steps), "image_bit_identical_at_8spp": same}
        if not args.no_cpu_baseline and world == 1:             # the CPU leg is timed at N = 1 only
            r.set_flags(0)                                      # production kernels for the parity render
            out["cpu_baseline"], out["parity"] = cpu_baseline(hrt, scene, args.cpu_seconds, r)
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
