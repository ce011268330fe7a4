"""Schedule of the dataflow SGS sweeps (nalu_amd/csrc/sgs_flow.hpp) replayed on the CPU: chunks in dispatch
order, steps in order, in-chunk values through a ring of the size the kernel uses, values of other chunks
only from rows marked visible -- against the sequential sweeps of Ifpack2's SGS
(src/LinearSolverConfig.C:71-75), bit for bit, on lattices (tiled and contiguous chunks) and random graphs.
Host logic only: no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_schedule_replays_the_sequential_sweeps(tmp_path):
    exe = str(tmp_path / "flowcheck")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I", os.path.join(ROOT, "nalu_amd", "csrc"),
                    os.path.join(ROOT, "scripts", "flowcheck.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "all schedules replay the sequential sweeps bit for bit" in out, out
    assert out.count("tiled chunks") >= 6 and out.count("contiguous chunks") >= 6, out
    assert "are not consecutive" in out, out  # a row with unsorted columns is refused, not mis-scheduled
