"""The reference's UnitTests/{wjacobi,gseidel,sor,vcycle,twogrid}Test.py in one script: same calls, the expected
values from their comments printed next to the result."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from MGCMTSolver import MGCMTSolver  # noqa: E402
from MGCMTStencilMaker import MGCMTStencilMaker  # noqa: E402

solver = MGCMTSolver()
stencil_maker = MGCMTStencilMaker()
gridsize = 2 ** 4
A = stencil_maker.laplacian(gridsize)
f = np.zeros((gridsize, 1))
for name, fn, expected in (("wjacobi", lambda x: solver.wjacobi(x, f, A, nu=4), 2.94959),
                           ("gseidel", lambda x: solver.gseidel(x, f, A), 1.88358),
                           ("sor", lambda x: solver.sor(x, f, A, nu=4, omega=2. / 3.), 2.63327)):
    x = np.ones((gridsize, 1))
    for _ in range(5):
        x = fn(x)
    print(name, np.linalg.norm(x), "# Expected value is", expected)
print("vcycle", np.linalg.norm(solver.vcycle(np.ones((gridsize, 1)), f, A, stencil_maker, nu1=4, nu2=4)), "# Expected value is 0.17756")
print("twogrid", np.linalg.norm(solver.twogrid(np.ones((gridsize, 1)), f, A, stencil_maker, nu1=4, nu2=4)), "# Expected value is 0.04979")
