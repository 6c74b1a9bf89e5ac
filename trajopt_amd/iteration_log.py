"""Host-side per-iteration log fed by the optimizer callbacks.

The reference prints / writes, per SQP iteration, the exact cost values and constraint violations before and after the
step together with the loop variables (trajopt_sco/src/optimizers.cpp:428-647: BasicTrustRegionSQPResults::print,
writeSolver / writeVars / writeCosts / writeConstraints into trajopt_vars.log, trajopt_costs.log, ...).  The persistent
kernel keeps the model ("approx") values on the device, so this log carries the exact values (oldexact / dexact of the
reference's table), the loop variables and the counters; `dapprox` / `ratio` are not available from the host.
Pure host code: it only consumes what BatchedTrustRegionSQP.addCallback delivers."""
import csv
import io
from typing import List, Optional, Sequence


class IterationLog:
    """callback object: opt.addCallback(IterationLog(cost_names, cnt_names))"""

    def __init__(self, cost_names: Optional[Sequence[str]] = None, cnt_names: Optional[Sequence[str]] = None):
        self.cost_names = list(cost_names) if cost_names is not None else None
        self.cnt_names = list(cnt_names) if cnt_names is not None else None
        self.rows: List[dict] = []
        self._last = {}

    def __call__(self, seed: int, r: dict):
        prev = self._last.get(seed)
        row = dict(seed=seed, merit_increases=r["merit_increases"], sqp_iter=r["sqp_iter"], trust_box_size=r["trust_box_size"],
                   n_qp_solves=r["n_qp_solves"], n_func_evals=r["n_func_evals"], status=r["status"],
                   cost_vals=[float(v) for v in r["cost_vals"]], cnt_viols=[float(v) for v in r["cnt_viols"]],
                   dexact_costs=None if prev is None else [a - float(b) for a, b in zip(prev["cost_vals"], r["cost_vals"])],
                   dexact_cnts=None if prev is None else [a - float(b) for a, b in zip(prev["cnt_viols"], r["cnt_viols"])])
        self.rows.append(row)
        self._last[seed] = row

    def _names(self, prefix, given, n):
        return list(given) if given is not None and len(given) == n else [f"{prefix}_{i}" for i in range(n)]

    def format_table(self, seed: int = 0) -> str:
        """one block per callback of `seed`, laid out like BasicTrustRegionSQPResults::print (exact columns only)"""
        out = io.StringIO()
        for row in (r for r in self.rows if r["seed"] == seed):
            out.write("| %s |\n" % ("=" * 76))
            out.write("| merit increases %d | SQP iteration %d | trust box %.3e | QP solves %d | status %d\n" %
                      (row["merit_increases"], row["sqp_iter"], row["trust_box_size"], row["n_qp_solves"], row["status"]))
            out.write("| %10s | %10s | %-30s\n" % ("exact", "dexact", "COSTS"))
            names = self._names("cost", self.cost_names, len(row["cost_vals"]))
            for i, v in enumerate(row["cost_vals"]):
                d = row["dexact_costs"][i] if row["dexact_costs"] is not None else float("nan")
                out.write("| %10.3e | %10.3e | %-30s\n" % (v, d, names[i]))
            out.write("| %10s | %10s | %-30s\n" % ("violation", "dexact", "CONSTRAINTS"))
            names = self._names("cnt", self.cnt_names, len(row["cnt_viols"]))
            for i, v in enumerate(row["cnt_viols"]):
                d = row["dexact_cnts"][i] if row["dexact_cnts"] is not None else float("nan")
                out.write("| %10.3e | %10.3e | %-30s\n" % (v, d, names[i]))
            out.write("| %10.3e | %10s | %-30s\n" % (sum(row["cost_vals"]), "", "TOTAL COST"))
        return out.getvalue()

    def write_csv(self, stream) -> None:
        """one line per callback: loop variables, counters, exact values (the union of the reference's solver / costs /
        constraints logs, minus the model values)"""
        n_c = max((len(r["cost_vals"]) for r in self.rows), default=0)
        n_v = max((len(r["cnt_viols"]) for r in self.rows), default=0)
        w = csv.writer(stream)
        w.writerow(["seed", "merit_increases", "sqp_iter", "trust_box_size", "n_qp_solves", "n_func_evals", "status"] +
                   self._names("cost", self.cost_names, n_c) + self._names("cnt", self.cnt_names, n_v))
        for r in self.rows:
            w.writerow([r["seed"], r["merit_increases"], r["sqp_iter"], "%e" % r["trust_box_size"], r["n_qp_solves"], r["n_func_evals"],
                        r["status"]] + ["%e" % v for v in r["cost_vals"]] + ["%e" % v for v in r["cnt_viols"]])
