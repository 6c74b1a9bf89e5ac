"""Host-side per-iteration log fed by the optimizer callbacks.

The reference prints / writes, per SQP iteration, the exact cost values and constraint violations before and after the
step together with the loop variables (trajopt_sco/src/optimizers.cpp:428-647: BasicTrustRegionSQPResults::print,
writeSolver / writeVars / writeCosts / writeConstraints into trajopt_vars.log, trajopt_costs.log, ...).  IterationLog
carries the loop variables, counters and exact values seen by the per-iteration callbacks (addCallback); StepTable carries
the reference's whole table - oldexact, new_exact, new_approx, dapprox, dexact, ratio per cost and constraint, the merit
coefficients and the merits - from the device-side record of every trust-region evaluation (addStepCallback,
tmx_sqp_step_log).  Pure host code: it only consumes what the callbacks deliver."""
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


class StepTable:
    """callback object for BatchedTrustRegionSQP.addStepCallback: one BasicTrustRegionSQPResults per trust-region evaluation"""

    def __init__(self, cost_names: Optional[Sequence[str]] = None, cnt_names: Optional[Sequence[str]] = None):
        self.cost_names = list(cost_names) if cost_names is not None else None
        self.cnt_names = list(cnt_names) if cnt_names is not None else None
        self.steps: List[tuple] = []   # (seed, step dict)

    def __call__(self, seed: int, step: dict):
        self.steps.append((seed, step))

    def of(self, seed: int) -> List[dict]:
        return [s for (b, s) in self.steps if b == seed]

    @staticmethod
    def format_step(step: dict, cost_names=None, cnt_names=None) -> str:
        """BasicTrustRegionSQPResults::print (optimizers.cpp:428-531): same columns, formats and '----------' placeholders"""
        out = io.StringIO()
        bar, dash, ph = "=" * 88, "-" * 88, "----------"
        nm = lambda names, i, p: names[i] if names is not None and i < len(names) else f"{p}_{i}"
        out.write("\n| %s |\n" % bar)
        out.write("| %10s | %10s | %10s | %10s | %10s | %10s | %10s |\n" % ("merit", "oldexact", "new_exact", "new_approx", "dapprox", "dexact", "ratio"))
        out.write("| %s | COSTS\n" % dash)
        for i in range(len(step["old_cost_vals"])):
            o, n, m = float(step["old_cost_vals"][i]), float(step["new_cost_vals"][i]), float(step["model_cost_vals"][i])
            da, de = o - m, o - n
            ratio = "%10.3e" % (de / da) if abs(da) > 1e-8 else "%10s" % ph
            out.write("| %10s | %10.3e | %10.3e | %10.3e | %10.3e | %10.3e | %s | %-15s \n" % (ph, o, n, m, da, de, ratio, nm(cost_names, i, "cost")))
        out.write("| %s |\n" % bar)
        out.write("| %10s | %10.3e | %10.3e | %10.3e | %10s | %10s | %10s | SUM COSTS\n" %
                  (ph, sum(step["old_cost_vals"]), sum(step["new_cost_vals"]), sum(step["model_cost_vals"]), ph, ph, ph))
        out.write("| %s |\n" % bar)
        if len(step["old_cnt_viols"]):
            out.write("| %s | CONSTRAINTS\n" % dash)
            for i in range(len(step["old_cnt_viols"])):
                o, n, m = float(step["old_cnt_viols"][i]), float(step["new_cnt_viols"][i]), float(step["model_cnt_viols"][i])
                mc = float(step["merit_error_coeffs"][i])
                da, de = o - m, o - n
                ratio = "%10.3e" % (de / da) if abs(da) > 1e-8 else "%10s" % ph
                out.write("| %10.3e | %10.3e | %10.3e | %10.3e | %10.3e | %10.3e | %s | %-15s \n" %
                          (mc, mc * o, mc * n, mc * m, mc * da, mc * de, ratio, nm(cnt_names, i, "cnt")))
        out.write("| %s |\n" % bar)
        out.write("| %10s | %10.3e | %10.3e | %10.3e | %10s | %10s | %10s | SUM CONSTRAINTS (WITHOUT MERIT) \n" %
                  (ph, sum(step["old_cnt_viols"]), sum(step["new_cnt_viols"]), sum(step["model_cnt_viols"]), ph, ph, ph))
        out.write("| %s |\n" % bar)
        out.write("| %10s | %10.3e | %10.3e | %10s | %10.3e | %10.3e | %10.3e | TOTAL = SUM COSTS + SUM CONSTRAINTS (WITH MERIT)\n" %
                  (ph, step["old_merit"], step["new_merit"], ph, step["approx_merit_improve"], step["exact_merit_improve"], step["merit_improve_ratio"]))
        out.write("| %s |\n" % bar)
        return out.getvalue()

    def format_table(self, seed: int = 0) -> str:
        return "".join(self.format_step(s, self.cost_names, self.cnt_names) for s in self.of(seed))
