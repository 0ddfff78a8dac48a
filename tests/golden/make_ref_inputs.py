"""Inputs of the reference harness (oracle/ref_harness/ref_main.cpp, scripts/build_ref_oracle.sh): the seeded synthetic problems of the
parity tests as flat dumps under tests/golden/ref_inputs/.  Deterministic: the same seeds give the same bytes here and in the tests."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from dmsa_lidar_slam_amd import dump, synth  # noqa: E402

CASES = {
    "window_small": lambda: synth.window_problem(seed=21, scans=3, rings=16, az_steps=160, num_static=0),
    "window_static": lambda: synth.window_problem(seed=22, scans=3, rings=32, az_steps=192, num_static=4000),
    "keyframes_small": lambda: synth.keyframe_problem(seed=23, frames=6, rings=16, az_steps=128, arc=0.4),
}
ITERATIONS = {"window_small": 4, "window_static": 4, "keyframes_small": 3}


def main():
    out = os.path.join(HERE, "ref_inputs")
    os.makedirs(out, exist_ok=True)
    for name, make in CASES.items():
        prob = make()
        path = os.path.join(out, name + ".bin")
        (dump.write_keyframe_map if name.startswith("keyframes") else dump.write_window_problem)(path, prob)
        print(path, os.path.getsize(path) >> 10, "KiB")


if __name__ == "__main__":
    main()
