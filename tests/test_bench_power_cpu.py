"""bench.py's board-power sampler (amdgpu hwmon files) on a fake sysfs tree: the card is the one whose power follows the warm-up, the timed region's samples are
averaged, and a box that exposes no telemetry for this job's card yields a note instead of a number."""
import importlib.util
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _card(tmp_path, name, power_uw, freq_hz, cap_uw=1_400_000_000):
    d = tmp_path / name / "device" / "hwmon" / "hwmon0"
    d.mkdir(parents=True)
    (d / "power1_input").write_text(str(power_uw))
    (d / "freq1_input").write_text(str(freq_hz))
    (d / "power1_cap").write_text(str(cap_uw))
    return d


def test_board_power_follows_the_busy_card(tmp_path, monkeypatch):
    bench = _bench()
    idle, mine = _card(tmp_path, "card0", 250_000_000, 95_000_000), _card(tmp_path, "card8", 245_000_000, 157_000_000)
    import glob as _glob

    real = _glob.glob
    monkeypatch.setattr(_glob, "glob", lambda pat: [str(idle / "power1_input"), str(mine / "power1_input")] if "hwmon" in pat else real(pat))
    bp = bench.BoardPower()
    (mine / "power1_input").write_text("1360000000")  # the warm-up steps load this job's card
    (mine / "freq1_input").write_text("1960000000")
    time.sleep(0.08)
    bp.choose()
    bp.mark()
    time.sleep(0.12)
    out = bp.result()
    assert bp.chosen == str(mine) and out["cap_W"] == 1400.0
    assert abs(out["mean_W"] - 1360.0) < 1e-6 and out["sclk_mean_MHz"] == 1960 and out["samples"] >= 3


def test_board_power_without_visible_telemetry(tmp_path, monkeypatch):
    bench = _bench()
    other = _card(tmp_path, "card3", 300_000_000, 95_000_000)
    import glob as _glob

    real = _glob.glob
    monkeypatch.setattr(_glob, "glob", lambda pat: [str(other / "power1_input")] if "hwmon" in pat else real(pat))
    bp = bench.BoardPower()
    time.sleep(0.05)
    bp.choose()  # nothing rose by 150 W: not this job's card
    bp.mark()
    out = bp.result()
    assert out["mean_W"] is None and "telemetry" in out["note"]
