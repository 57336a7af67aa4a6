"""Host-side behaviour of bench.py that needs no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_must_match_gpus():
    """A launcher that started another number of ranks than --gpus names ends the run with a non-zero status before
    anything is measured, instead of a line that claims --gpus."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "started 1 rank" in r.stderr and "{" not in r.stdout
