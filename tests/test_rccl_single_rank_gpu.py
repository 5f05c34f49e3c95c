"""GPU: the trajectory all-gather through RCCL itself (backend "nccl") with the one rank a 1-GPU box allows.

The multi-rank layout is rehearsed over gloo (tests/test_distributed_gloo.py); what gloo cannot show is the RCCL side of
the plumbing: process-group init on the env's device, the collective on its own stream, the stream-level wait before a
chunk is refilled, device tensors in and out.  world_size = 1 exercises exactly that (RCCL refuses two ranks on one
GPU, and the N = 8 run is the driver's)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["FE_REPO"])
import finenvs_amd
from finenvs_amd.data import synthetic
from finenvs_amd.trajectory import TrajectoryBuffer

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
N, A, W, T = 4096, 2, 8, 6
prices, day_id, _ = synthetic.synthetic_series(6, A, 60, 1234)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=3)
traj = TrajectoryBuffer(T, N, A, states=True)
g = torch.Generator(device="cuda").manual_seed(1)
obs = env.reset()
traj.begin(env)
for chunk in range(3):
    seen = []
    for t in range(T):
        seen.append(obs.clone())
        a, r, d = traj.next_slot()
        a.copy_(torch.rand((N, A), generator=g, device="cuda") * 2 - 1)
        obs, *_ = env.step(a, rewards_out=r, dones_out=d, descriptors_out=traj.state_slot())
    want = (traj.actions.clone(), traj.rewards.clone(), traj.dones.clone(), traj.obs_src.clone(), traj.obs_pos.clone())
    traj.all_gather_async()          # RCCL, on its own stream; the next chunk fills meanwhile
    got = traj.wait_gathered(with_states=True)
    for name, w, gt in zip(("actions", "rewards", "dones", "obs_src", "obs_pos"), want, got):
        assert gt.shape[0] == 1 and torch.equal(gt[0], w), name
    # the gathered descriptors render the states the policy saw
    for t in (0, T - 1):
        assert torch.equal(env.render(got[3][0, t], got[4][0, t]), seen[t]), f"state {t}"
    assert torch.equal(env.render(got[3][0, T], got[4][0, T]), obs)
traj.drain()
blocking = TrajectoryBuffer(T, N, A)
for t in range(T):
    blocking.store(torch.full((N, A), float(t), device="cuda"), torch.full((N,), 2.0 * t, dtype=torch.float64, device="cuda"),
                   torch.zeros((N,), dtype=torch.int32, device="cuda"))
a, r, d, packed = blocking.all_gather()
assert torch.equal(a[0], blocking.actions) and torch.equal(r[0], blocking.rewards) and packed.shape == (1, blocking._nbytes)
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL single-rank all-gather ok")
'''


def test_trajectory_all_gather_through_rccl_one_rank(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               FE_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "RCCL single-rank all-gather ok" in out.stdout


CAPTURE_WORKER = r'''
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["FE_REPO"])
import finenvs_amd
from finenvs_amd.data import synthetic
from finenvs_amd.rollout import GraphedRollout

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
N, A, W, K = 2048, 1, 8, 8
prices, day_id, _ = synthetic.synthetic_series(6, A, 60, 1234)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=3, obs_buffers=2)
acts = [(torch.rand((N, A), device="cuda") * 2 - 1) for _ in range(K)]
x = torch.ones((4,), device="cuda")
t0 = time.perf_counter()
captures = 0
# A collective leaves work for the process group's watchdog thread to poll (an event query every ~100 ms until it has seen the work
# complete); graph captures go on meanwhile for more than a second: some poll falls into some capture.
while time.perf_counter() - t0 < 1.5:
    dist.all_reduce(x, async_op=True)
    roll = GraphedRollout(env, lambda obs, k: acts[k], K, warmup=0)
    roll.run()
    captures += 1
torch.cuda.synchronize()
dist.destroy_process_group()
print(f"captured {captures} graphs beside a polling RCCL watchdog: ok")
'''


def test_graph_capture_beside_the_process_groups_watchdog_thread(tmp_path):
    """GraphedRollout captures in THREAD-LOCAL mode.  ProcessGroupNCCL's watchdog thread polls its collectives' events from another
    thread; under torch's default global capture mode such a poll during a capture fails with "operation not permitted when stream
    is capturing", the watchdog rethrows and the process is std::terminate()d -- which is how one of ~10 rehearsals of bench.py's
    N > 1 path died in round 6 (exit code -6, after the headline, before its line).  Collectives and captures interleaved for 1.5 s
    reproduce the collision; with the thread-local mode the worker must come through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "capture_worker.py"
    script.write_text(CAPTURE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               FE_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout[-1000:] + out.stderr[-4000:])
    assert "beside a polling RCCL watchdog: ok" in out.stdout
