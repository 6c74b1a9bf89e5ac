"""The oracle pinned against the reference's own portable known-answer tests:
trajopt_sco/test/solver-utils-unit.cpp:19-244, solver-interface-unit.cpp:21-31,136-237,
small-problems-unit.cpp:48-172 — replayed by oracle/kat.cpp."""
import pytest

EXPECTED = [
    "solver_utils.exprToEigen.affine_vector", "solver_utils.exprToEigen.u_is_minus_constant", "solver_utils.exprToEigen.A",
    "solver_utils.exprToEigen.q", "solver_utils.exprToEigen.Q", "solver_utils.exprToEigen.Q_halved",
    "solver_utils.exprToEigen.Q_zero_dropped", "solver_utils.exprToEigen.Q_zero_dropped_halved",
    "solver_utils.exprToEigen.Q_force_diagonal", "solver_utils.exprToEigen.Q_force_diagonal_halved",
    "solver_utils.eigenToCSC.values", "solver_utils.eigenToCSC.rows", "solver_utils.eigenToCSC.colptr",
    "solver_utils.eigenToCSC.two_entries", "solver_utils.eigenToCSC.one_entry", "solver_utils.eigenToCSC_upper_triangular",
    "SolverInterface.simplify2", "SolverInterface.ExprMult_test2", "SolverInterface.ExprMult_test3",
    "SolverInterface.remove_var_renumbers", "OSQP.demo_qp", "SQP.QuadraticSeparable", "SQP.QuadraticNonseparable",
    "SQP.TP1", "SQP.TP3", "SQP.TP6", "SQP.TP7",
]


@pytest.fixture(scope="module")
def kat_lines(orc):
    rc, out = orc.run_kat()
    lines = {}
    for ln in out.splitlines():
        parts = ln.split()
        if len(parts) >= 2 and parts[0] in ("PASS", "FAIL"):
            lines[parts[1]] = (parts[0], ln)
    return rc, lines


@pytest.mark.parametrize("name", EXPECTED)
def test_reference_kat(kat_lines, name):
    rc, lines = kat_lines
    assert name in lines, f"KAT {name} did not run"
    assert lines[name][0] == "PASS", lines[name][1]


def test_kat_binary_exit_code(kat_lines):
    assert kat_lines[0] == 0
