"""GPU box: throughput of the fused K-step rollouts (in-kernel policy, observation never written to HBM):
linear window form, linear table form, MLP head and LSTM head on the matrix cores.

    python tools/fused_bench.py <config> [form ...]      forms: window table mlp32 mlp64 mlp128 lstm32 lstm64 lstm128
    FUSED_W=4 overrides the config's window (the reference's LSTM scripts use num_intervals=4), FUSED_N the env count,
    FUSED_K the steps per launch, FUSED_TILES a list of rollout tile overrides, FUSED_LIB the tag of an experiment build.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402
from finenvs_amd.rollout import FusedLinearRollout, FusedLSTMRollout, FusedMLPRollout  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
forms = sys.argv[2:] or ["window", "table", "mlp32", "mlp64"]
name, N, A, W = CONFIGS[cfg]
W = int(os.environ.get("FUSED_W", W))
N = int(os.environ.get("FUSED_N", N))  # e.g. an evaluation over a few thousand trading days
prices, day_id, _ = make_series(A)
K = int(os.environ.get("FUSED_K", "32"))
tiles = [int(x) for x in os.environ.get("FUSED_TILES", "0").split(",")]
g = torch.Generator().manual_seed(0)
native = None
if os.environ.get("FUSED_LIB"):  # an experiment build (finenvs_amd.csrc.build.build_variant) by its tag
    from finenvs_amd import _lib
    native = _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", f"libfinenvs_amd.{os.environ['FUSED_LIB']}.so"))
for form in forms:
    for eb in tiles:
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", obs_buffers=1,
                                        **({"_native": native} if native is not None else {}))
        if eb:
            env.set_launch(0, 0, eb)
        if form.startswith("torchlstm"):
            # the unfused loop of the reference's scripts on this GPU: torch's own nn.LSTM (MIOpen / rocBLAS) on
            # states.float(), then env.step -- what FusedLSTMRollout replaces.  f64 observations are written and re-read.
            H = int(form[9:])
            torch.manual_seed(0)
            lstm, lin = torch.nn.LSTM(5, H, batch_first=True).cuda(), torch.nn.Linear(H, 1).cuda()
            states = env.reset()

            def loop(k):
                global states
                with torch.no_grad():
                    for _ in range(k):
                        x = states.float()
                        acts = torch.stack([torch.tanh(lin(lstm(x[:, :, 5 * a:5 * a + 5])[0][:, -1, :])).squeeze(1) for a in range(A)], 1)
                        states, _, _, _ = env.step(acts)

            loop(K)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            loop(3 * K)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / (3 * K)
            print(f"config {cfg} W {W} {form:12s} (unfused: torch nn.LSTM + env.step): {ms * 1e3:9.2f} us/step  {N / ms / 1e6:8.3f} G env-steps/s", flush=True)
            del env
            continue
        if form.startswith("lstm"):
            H = int(form[4:])
            torch.manual_seed(0)
            lstm, lin = torch.nn.LSTM(5, H, batch_first=True), torch.nn.Linear(H, 1)
            with torch.no_grad():
                lstm.weight_ih_l0[:, :4].mul_(6.0 * H ** 0.5)
            roll = FusedLSTMRollout.from_modules(env, lstm, lin)
            if os.environ.get("FUSED_SPLIT"):  # 1 / 0: force the per-time-step (split) or the fused large-H path
                roll.split = os.environ["FUSED_SPLIT"] == "1"
            flop = 2.0 * N * A * 4 * H * (8 * W + H * (W - 1))  # gate contractions as executed, per step
        elif form.startswith("mlp"):
            H = int(form[3:])
            W1 = torch.randn((5 * W, H), generator=g) * (8.0 / W ** 0.5)
            roll = FusedMLPRollout(env, W1, torch.randn(H, generator=g) * 0.3, torch.randn(H, generator=g) / H ** 0.5, 0.0)
            flop = 2.0 * N * A * (4 * W) * H  # first layer, per step
        else:
            roll = FusedLinearRollout(env, torch.randn((W, 5), dtype=torch.float64, generator=g) * 2, 0.0, form=form)
            flop = 0.0
        roll.run(K, record_actions=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        reps = 5
        for _ in range(reps):
            roll.run(K, record_actions=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (reps * K)
        extra = f"  MFMA part {flop / ms / 1e9:7.1f} TFLOP/s f32 (peak 157)" if flop else ""
        print(f"config {cfg} W {W} {form:7s} tile {eb or 'auto':>4}: {ms * 1e3:9.2f} us/step  {N / ms / 1e6:8.3f} G env-steps/s"
              f"  {N * A / ms / 1e6:8.3f} G account-steps/s{extra}", flush=True)
        del env, roll
