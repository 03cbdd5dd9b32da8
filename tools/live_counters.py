"""Live rocprofv3 counter passes as CHILD processes of a measuring program (bench.py).

The guide (MI355X_MICROARCH.md, section HBM / rocprofv3 PMC slots) wants FETCH_SIZE and WRITE_SIZE in separate --pmc
passes, never together with runtime / system tracing; a running program cannot attach counters to itself, so bench.py
starts small stand-alone replays of the kernel it has just timed (tools/trace_only.py, tools/sort_bench.py) under
`rocprofv3 --kernel-trace --pmc ...` and reads their CSVs.  The profiled program is given to rocprofv3 directly
(`-- python3 script.py`): no shell, no env wrapper, no re-exec between the profiler's preloaded library and HIP.
Everything here returns None on any failure (no rocprofv3, a timeout, an unreadable CSV): the bench line then says
`"traffic": null` instead of replaying numbers from an earlier run.
"""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rocprof():
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


def run_pass(counters, script_args, timeout_s=240):
    """One rocprofv3 pass.  Returns [(kernel_name, dispatch_id, counter_name, value, start_ns, end_ns)] or None."""
    exe = rocprof()
    if exe is None:
        return None
    out = tempfile.mkdtemp(prefix="lbvh_pmc_", dir="/tmp")
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [exe, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out, "--", sys.executable] + list(script_args)
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        rows = []
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                rows.append((r["Kernel_Name"], int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"]),
                             int(r.get("Start_Timestamp", 0) or 0), int(r.get("End_Timestamp", 0) or 0)))
        return rows or None
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def per_kernel(rows, kernel_substr, pick="last"):
    """{counter: value} of the chosen dispatch(es) of the kernels whose name contains kernel_substr.
    pick = "last": the last dispatch (the warmed-up, cost-ordered one); "mean": mean over all dispatches."""
    if not rows:
        return None
    sel = [r for r in rows if kernel_substr in r[0]]
    if not sel:
        return None
    out = {}
    if pick == "last":
        last = max(r[1] for r in sel)
        for r in sel:
            if r[1] == last:
                out[r[2]] = out.get(r[2], 0.0) + r[3]
    else:
        n = len({r[1] for r in sel})
        for r in sel:
            out[r[2]] = out.get(r[2], 0.0) + r[3] / n
    return out


def hbm_traffic(script_args, kernel_substr, pick="last"):
    """HBM-side bytes per launch of one kernel: FETCH_SIZE and WRITE_SIZE in their own passes (both in KiB); FETCH_SIZE
    doubled as the gfx950 note prescribes (128-B read requests tallied at 64 B)."""
    f = per_kernel(run_pass(["FETCH_SIZE"], script_args), kernel_substr, pick)
    w = per_kernel(run_pass(["WRITE_SIZE"], script_args), kernel_substr, pick)
    if not f or not w or "FETCH_SIZE" not in f or "WRITE_SIZE" not in w:
        return None
    fetch = 2.0 * f["FETCH_SIZE"] * 1024.0
    write = w["WRITE_SIZE"] * 1024.0
    return {"bytes": round(fetch + write), "fetch_bytes_x2": round(fetch), "write_bytes": round(write),
            "source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run (FETCH_SIZE x2, gfx950)"}


SQ_GROUP = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY",
            "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]


def issue_counters(script_args, kernel_substr, pick="last"):
    return per_kernel(run_pass(SQ_GROUP, script_args), kernel_substr, pick)
