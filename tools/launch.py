#!/usr/bin/env python3
"""One process per GPU for any program that reads RANK / LOCAL_RANK / WORLD_SIZE (+ OFFT_ID_FILE for the C harness):

  python tools/launch.py -n 8 -- bin/run-fft -N 1024 -n 1024 -L 1024 -r 5 -d 1 -v

the MPI-free counterpart of the reference's `mpiexec -n 8 ./run-fft ...` (job-test.sh:9-13).  Same watchdog as
`bench.py --gpus N`: children in their own sessions, the others are stopped when one fails, a wall-clock limit, the
failing rank's tail on stderr, rank 0's stdout relayed."""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # launcher only


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", "--nproc", type=int, required=True)
    ap.add_argument("--launch-timeout", type=float, default=900.0)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    if not cmd:
        ap.error("no command")
    os.environ["OFFT_ID_FILE"] = os.path.join(tempfile.mkdtemp(prefix="offt_id_"), "rccl_id")
    os.environ["OFFT_RUN_NONCE"] = "%d-%d" % (os.getpid(), int(time.time() * 1e6))  # run-fft accepts only this run's id file

    class Args:
        gpus = a.nproc
        launch_timeout = a.launch_timeout
    sys.exit(bench.launch(Args, cmd[1:], script=cmd[0], interpreter=None if os.access(cmd[0], os.X_OK) and not cmd[0].endswith(".py") else sys.executable))


if __name__ == "__main__":
    main()
