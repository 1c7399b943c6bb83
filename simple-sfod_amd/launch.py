"""One process per GPU, started by the entry point itself -- the role of ``detectron2.engine.launch`` in the
reference's ``train_net_mt.py:90-101`` (``launch(main, args.num_gpus, num_machines=1, dist_url=..., args=(args,))``).

``python train_net_mt.py --num-gpus N ...`` / ``python bench.py --gpus N ...`` re-run their own script as N ranks
through ``python -m torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment; rendezvous on
127.0.0.1) in a CHILD process and exit with its code.  The parent never touches the GPU: this module imports the
standard library only, and the callers invoke it before importing torch (replacing or forking a process that has
initialised HIP is not safe on this platform).
"""
import os
import socket
import subprocess
import sys


def under_launcher():
    """True inside a rank started by torch.distributed.run (or any launcher that exports the rendezvous variables)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(script, argv, nproc, port=None):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(argv)


def launch(script, argv, nproc, env=None):
    """Run ``script argv`` as ``nproc`` ranks; -> the launcher's exit code (0 only if every rank succeeded)."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver stack
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(int(nproc), 1))))
    return subprocess.call(launch_command(script, argv, nproc), env=e)
