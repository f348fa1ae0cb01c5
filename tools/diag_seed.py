#!/usr/bin/env python
"""Developer tool: IPM trajectories of one cfg-4 batch problem, HIP path vs the oracle on the same elimination order, iteration by
iteration (where do they part, and by how much).  usage: [HIPKKT_...] diag_seed.py <seed> [label]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clarabel_jl_amd  # noqa: F401
import julia_standin as cl
from clarabel_jl_amd import problems
from oracle.kkt_oracle import OracleKKTSolver
seed = int(sys.argv[1]); label = sys.argv[2] if len(sys.argv) > 2 else ""
P, q, A, b, cones = problems.batch_problem(seed)
sg = cl.Solver(P, q, A, b, cones, cl.Settings()); sg.trace = []
solg = sg.solve()
ks = sg.kktsystem.kktsolver
perm = ks.h.perm()
sc = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: OracleKKTSolver(*a, ordering=perm)); sc.trace = []
solc = sc.solve()
print(f"DIAG {label} seed {seed} n={A.shape[1]} m={A.shape[0]} N={ks.h.N} nnzL={ks.h.nnzL} fronts={ks.h.counters()['fronts']} status {solg.status}/{solc.status} it {solg.iterations}/{solc.iterations} "
      f"ir_total hip {ks.total_ir_steps} oracle {sc.kktsystem.kktsolver.total_ir_steps} timeouts {ks.h.counters()['sweep_timeouts']}")
for tg, tc in zip(sg.trace, sc.trace):
    print(f"DIAG {label} it {tg['iter']:2d} dcost {abs(tg['cost_primal']-tc['cost_primal'])/max(1,abs(tc['cost_primal'])):.2e} dres {max(abs(tg['res_primal']-tc['res_primal']),abs(tg['res_dual']-tc['res_dual'])):.2e} "
          f"alpha {tg['alpha']:.6f}/{tc['alpha']:.6f} mu {tg['mu']:.3e} res_p {tc['res_primal']:.2e} res_d {tc['res_dual']:.2e}")
