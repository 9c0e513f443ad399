"""One-off soak: random sharded cycles on ONE GPU (gloo ranks sharing the device, host-staged messages) against the
single-plan cycle.  usage: soak_sharded_gpu.py [cases] [seed]"""
import os, socket, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
child = os.path.join(ROOT, "tests", "sharded_gloo_gpu.py")
for c in range(cases):
    world = int(rng.choice([2, 4]))
    g = int(rng.choice([1024, 2048, 4096]))
    kind_name = str(rng.choice(["wjacobi", "rb"]))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [subprocess.Popen([sys.executable, child, str(r), str(world), port, str(g), kind_name, tmp], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = [p.communicate(timeout=600)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        got = np.concatenate([np.load(os.path.join(tmp, "part%d.npy" % r)) for r in range(world)])
    kind, omega = (_lib.WJACOBI, 2. / 3.) if kind_name == "wjacobi" else (_lib.GS_MC, 1.0)
    r2 = np.random.RandomState(9)
    f, v0 = r2.rand(g * g), r2.rand(g * g)
    p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
    p.set_shifts([0.3]); p.upload(0, _lib.SLOT_F, 0, f); p.upload(0, _lib.SLOT_V, 0, v0)
    for _ in range(2): p.vcycle(2, 2, kind, omega=omega, nu_coarse=2)
    want = p.download(0, _lib.SLOT_V, 0); p.close()
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    print("case", c, "world", world, "grid", g, kind_name, "rel err", err, flush=True)
    assert err < 1e-12
print("SOAK_OK")
