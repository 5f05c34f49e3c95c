"""The reference's own env tests (tests/unit/test_time_series_env.py:7-40 and
tests/integration/test_SPY_training.py:10-34), restated against the drop-in class: same
construction from CSV files by instrument name + dataset key, same type/shape assertions,
1000 consecutive steps.  The reference's data files do not travel, so the three instruments
are synthetic CSVs written in its row format."""
from typing import Dict, Tuple

import numpy as np
import pytest
import torch

from finenvs_amd.data import synthetic
from tests.helpers import assert_bits, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def envs(tmp_path_factory):
    from finenvs_amd import TimeSeriesEnv

    root = tmp_path_factory.mktemp("fin") / "data"
    out = []
    for name, days, bars, seed, drop in (("IBM", 4, 390, 3, 0.02), ("OIH", 12, 120, 4, 0.3), ("SPY", 9, 390, 5, 0.0)):
        prices, day_id, minute = synthetic.synthetic_series(days, 1, bars, seed, drop)
        synthetic.write_csv(str(root / name / "dummy.csv"), prices, day_id, minute, 0, premarket_rows=4)
        out.append(TimeSeriesEnv(str(root / name), "dummy", num_intervals=60))
    return out


def step_helper(env):
    num_envs = env.num_envs
    actions = torch.rand((num_envs, 1), device=env.device) * 2 - 1
    step_info: Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Dict] = env.step(actions)
    assert isinstance(step_info, tuple)
    (next_states, rewards, dones, info) = step_info
    assert isinstance(next_states, torch.Tensor)
    assert isinstance(rewards, torch.Tensor)
    assert isinstance(dones, torch.Tensor)
    assert isinstance(info, dict)
    return step_info


def test_should_reset_envs(envs):
    for env in envs:
        assert isinstance(env.reset(), torch.Tensor)


def test_should_step_envs(envs):
    for env in envs:
        obs, rew, done, _ = step_helper(env)
        assert obs.shape == (env.num_envs, env.num_intervals, env.num_obs) and obs.dtype == torch.float64
        assert rew.shape == (env.num_envs,) and rew.dtype == torch.float64
        assert done.shape == (env.num_envs,) and done.dtype == torch.int32


def test_should_step_envs_1000_times(envs):
    for _ in range(1000):
        for env in envs:
            step_helper(env)
    for env in envs:
        assert bool(torch.isfinite(env.cash).all()) and int(env.env_spots.min()) >= 0


def _write_reference_style_csv(path, inst, prices, day_id, second):
    """A CSV in the row format of the reference's own fixture for `inst` (finenvs/data/README.md:7-9): IBM / OIH spell
    MM/DD/YYYY,HH:MM, SPY spells YYYY-MM-DD,HH:MM:SS and carries pre-market rows from 04:00 (dropped by the 09:30-15:59
    filter, TSE:90-91).  Prices are written with repr(): they parse back to the same doubles."""
    import os

    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        last = -1
        for row, d, sec in zip(prices, day_id, second):
            d, sec = int(d), int(sec)
            date = f"2022-04-{1 + d:02d}" if inst == "SPY" else f"{1 + d // 28:02d}/{1 + d % 28:02d}/1998"
            if inst == "SPY" and d != last:  # a few pre-market bars before each session
                for k in range(3):
                    f.write(f"{date},04:{k:02d}:00,1,1,1,1,100\n")
                last = d
            hh, mm, ss = sec // 3600, sec // 60 % 60, sec % 60
            time = f"{hh:02d}:{mm:02d}:{ss:02d}" if inst == "SPY" else f"{hh:02d}:{mm:02d}"
            f.write(f"{date},{time}," + ",".join(repr(float(x)) for x in row) + ",1000\n")


@pytest.mark.parametrize("inst,fixture,kw", [("IBM", "tables_ibm.npz", {}), ("SPY", "tables_spy.npz", {}),
                                             ("OIH", "tables_oih.npz", {"num_intervals": 32})])
def test_reference_datasets_through_the_constructor_the_reference_test_uses(tmp_path, inst, fixture, kw):
    """`TimeSeriesEnv("IBM", "dummy")` etc. as in the reference's unit test (tests/unit/test_time_series_env.py:10-14,
    default num_intervals = 390, TSE:19), on the market-hours rows of the reference's own datasets (from the
    fixtures: the files themselves do not travel) written back in each file's own date / time spelling: CSV ->
    native reader -> bounds -> device tables must equal what the reference built, then 1000 steps as its test does."""
    from finenvs_amd import TimeSeriesEnv

    g = load_golden(fixture)
    d = tmp_path / "data" / inst
    _write_reference_style_csv(str(d / "dummy.csv"), inst, g["series_prices"], g["series_day_id"], g["series_second"])
    env = TimeSeriesEnv(str(d), "dummy", **kw)
    assert env.num_intervals == int(g["W"]) and env.num_envs == g["ref_price_environments"].shape[0] + 1
    assert_bits(env.dataset.cpu().numpy(), g["ref_dataset"], "parsed market-hours frame")
    assert_bits(env.price_environments.cpu().numpy(), g["ref_price_environments"], "price tables")
    np.testing.assert_allclose(env.log_return_environments.cpu().numpy(), g["ref_log_return_environments"], rtol=1e-13, atol=1e-17)
    assert isinstance(env.reset(), torch.Tensor)
    for _ in range(1000):
        step_helper(env)
    assert bool(torch.isfinite(env.cash).all())


def test_env_built_from_csv_equals_the_reference_tables(tmp_path):
    """End to end: CSV file -> native reader -> bounds -> device transform/tables, against what the
    reference built from the very same CSV (tests/golden/tables_ragged.npz)."""
    from finenvs_amd import TimeSeriesEnv

    g = load_golden("tables_ragged.npz")
    prices, day_id, minute = synthetic.synthetic_series(7, 1, 40, 77, 0.10)
    d = tmp_path / "data" / "SYN"
    synthetic.write_csv(str(d / "dummy.csv"), prices, day_id, minute, 0, premarket_rows=2)
    env = TimeSeriesEnv(str(d), "dummy", num_intervals=int(g["W"]))
    D = g["ref_price_environments"].shape[0]
    assert env.num_envs == D + 1  # training mode adds the evaluation env (TSE:253-257)
    assert_bits(env.price_environments.cpu().numpy(), g["ref_price_environments"])
    np.testing.assert_allclose(env.log_return_environments.cpu().numpy(), g["ref_log_return_environments"],
                               rtol=1e-13, atol=1e-17)
    assert env.get_env_args()["env_name"] == str(d)
    with pytest.raises(Exception, match="dataset_key expected"):
        TimeSeriesEnv(str(d), "bogus")
    with pytest.raises(Exception, match="No file was found"):
        TimeSeriesEnv(str(d), "train")


def test_portfolio_env_from_several_instruments(tmp_path):
    from finenvs_amd import TimeSeriesEnv

    prices, day_id, minute = synthetic.synthetic_series(5, 3, 60, 9)
    names = []
    for a in range(3):
        d = tmp_path / "data" / f"AST{a}"
        synthetic.write_csv(str(d / "dummy.csv"), prices, day_id, minute, a)
        names.append(str(d))
    env = TimeSeriesEnv(names, "dummy", num_intervals=10, evaluate=True)
    assert env.num_assets == 3 and env.num_obs == 15 and env.num_acts == 3
    obs, rew, done, _ = env.step(torch.zeros((env.num_envs, 3), device=env.device))
    assert obs.shape == (env.num_envs, 10, 15)
    # each asset's sleeve of the joint tables equals the single-instrument env's tables
    single = TimeSeriesEnv(names[1], "dummy", num_intervals=10, evaluate=True)
    assert torch.equal(env.price_environments[:, :, 4:8], single.price_environments)
    assert torch.equal(env.log_return_environments[:, :, 4:8].nan_to_num(7.0), single.log_return_environments.nan_to_num(7.0))


def test_example_rollout_loop_eager_and_graphed_agree():
    """examples/time_series_rollout.py: the reference's loop shape with a stand-in LSTM policy;
    the hipGraph form must produce the very same trajectory as the eager loop."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "time_series_rollout.py")
    spec = importlib.util.spec_from_file_location("time_series_rollout", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    log_e, ret_e = mod.main(envs=512, window=4, iters=30, steps=16, graph=False, hidden=16, seed=1)
    log_g, ret_g = mod.main(envs=512, window=4, iters=30, steps=16, graph=True, hidden=16, seed=1)
    assert log_e["num_training_episodes"] == log_g["num_training_episodes"] > 0
    assert torch.equal(ret_e, ret_g)
