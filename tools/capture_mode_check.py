#!/usr/bin/env python3
"""GPU box: why GraphedRollout captures in THREAD-LOCAL mode (round 6).

    python tools/capture_mode_check.py

Runs the worker of tests/test_rccl_single_rank_gpu.py::test_graph_capture_beside_the_process_groups_watchdog_thread twice in child
processes -- RCCL process group of one rank, collectives and hipGraph captures interleaved for 1.5 s -- once as the product captures
(capture_error_mode="thread_local") and once with torch's default ("global") patched back in.  Under the global mode a poll of the
process group's watchdog thread (hipEventQuery on a collective's event, from ANOTHER thread) that falls into a capture fails with
"operation not permitted when stream is capturing"; the watchdog rethrows and the process is std::terminate()d (exit code -6).
That is how one in ~10 rehearsals of bench.py's N > 1 path died before its line was out."""
import os
import re
import socket
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(REPO, "tests", "test_rccl_single_rank_gpu.py")).read()
worker = re.search(r"CAPTURE_WORKER = r'''(.*?)'''", src, re.S).group(1)
FORCE_GLOBAL = '''
import torch
_orig = torch.cuda.graph.__init__
def _init(self, g, pool=None, stream=None, capture_error_mode="global"):
    _orig(self, g, pool=pool, stream=stream, capture_error_mode="global")   # whatever the caller asked for
torch.cuda.graph.__init__ = _init
'''
for label, prefix in (("thread_local (the product)", ""), ("global (torch's default, patched back in)", FORCE_GLOBAL)):
    path = f"/tmp/fe_capture_{'global' if prefix else 'local'}.py"
    open(path, "w").write(prefix + worker)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", FE_REPO=REPO)
    out = subprocess.run([sys.executable, path], env=env, capture_output=True, text=True, timeout=300)
    said = [ln for ln in out.stdout.splitlines() if "captured" in ln][-1:] or ["(no result line)"]
    why = [ln.split("] ", 1)[-1][:160] for ln in out.stderr.splitlines() if "capturing" in ln][:1]
    print(f"capture_error_mode = {label}: exit code {out.returncode}; {said[0]}" + (f"; stderr: {why[0]}" if why else ""))
