"""diagnostic: capture the factorisation of cfg 5 that fails in the cheap ("cone rows first") order -- the KKT values, the
permutation and the D it produced -- into gpurun_out/cfg5_fail.npz for an offline look with the oracle (tools/diag_cfg5_offline.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clarabel_jl_amd  # noqa: F401
import julia_standin as cl
from clarabel_jl_amd import problems
from clarabel_jl_amd.kktsolver import HipKKTSolver

P, q, A, b, specs = problems.sdp_blocks(seed=5)
state = dict(done=False, it=0)


class Spy(HipKKTSolver):
    def __init__(self, P_, A_, cones_, m_, n_, settings_, **kw):
        self._args = (P_, A_, m_, n_, settings_)
        super().__init__(P_, A_, cones_, m_, n_, settings_, **kw)

    def kktsolver_update(self, cones_):
        before = self.h.counters()["twin_refactors"]
        ok = super().kktsolver_update(cones_)
        state["it"] += 1
        if not state["done"] and self.h.counters()["twin_refactors"] > before:
            state["done"] = True
            D = self.h.debug_dump(5)
            kv = self.h.debug_dump(4)
            print(f"update {state['it']}: twin used; eps {self.diagonal_regularizer:.3e} nreg {self.last_nreg}; D of the failed attempt: "
                  f"nonfinite {np.sum(~np.isfinite(D))} first at {np.argmax(~np.isfinite(D))} of {len(D)}; |D| range "
                  f"{np.nanmin(np.abs(D)):.3e} .. {np.nanmax(np.abs(D[np.isfinite(D)])):.3e}; |K| max {np.abs(kv).max():.3e}")
            from oracle.kkt_oracle import OracleKKTSolver
            import time
            P_, A_, m_, n_, st_ = self._args
            t0 = time.time()
            o = OracleKKTSolver(P_, A_, cones_, m_, n_, st_, ordering=self.h.perm())
            ook = o.kktsolver_update(cones_)
            Do = o.k.factor_D()
            print(f"oracle in the same order: ok {ook} eps {o.diagonal_regularizer:.3e} nreg {o.k.L.oracle_kkt_nreg(o.k.h)} "
                  f"nonfinite {np.sum(~np.isfinite(Do))} ({time.time() - t0:.1f} s)")
            fin = np.isfinite(D) & np.isfinite(Do)
            rel = np.abs(D - Do) / np.maximum(np.abs(Do), 1e-300)
            bad = np.where(~fin | (rel > 1e-6))[0]
            first = self.h.debug_dump(10).astype(np.int64)
            level = self.h.debug_dump(11).astype(np.int64)
            rows = self.h.debug_dump(12).astype(np.int64)
            print("pivots that differ (rel > 1e-6 or non-finite):", len(bad), "of", len(D), "first", bad[:10])
            if len(bad):
                j = bad[0]
                s_ = int(np.searchsorted(first, j, side="right") - 1)
                print(f"first bad pivot {j}: supernode {s_} cols {first[s_]}..{first[s_ + 1]} level {level[s_]} rows {rows[s_]}; hip D {D[j]:.6e} oracle D {Do[j]:.6e}")
                lo = max(first[s_], j - 5)
                print("  hip   ", D[lo:j + 3]); print("  oracle", Do[lo:j + 3])
                print("  |oracle D| range in this supernode", np.abs(Do[first[s_]:first[s_ + 1]]).min(), np.abs(Do[first[s_]:first[s_ + 1]]).max())
            print("oracle |D| range", np.abs(Do).min(), np.abs(Do).max())
        return ok


st = cl.Settings()
solver = cl.Solver(P, q, A, b, specs, st, kktsolver_factory=lambda *a: Spy(*a))
sol = solver.solve()
print("status", sol.status, "iterations", sol.iterations, "counters", solver.kktsystem.kktsolver.h.counters())
