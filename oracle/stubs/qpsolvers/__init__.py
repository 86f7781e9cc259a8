"""Stand-in for ``qpsolvers``: Problem + solve_problem(solver="quadprog") backed by
oracle/qp_gi.py.  See oracle/stubs/README.md."""

from __future__ import annotations

import warnings
from dataclasses import dataclass
from typing import Optional

import numpy as np

from oracle import qp_gi


class ProblemError(Exception):
    pass


@dataclass
class Problem:
    P: np.ndarray
    q: np.ndarray
    G: Optional[np.ndarray] = None
    h: Optional[np.ndarray] = None

    def unpack(self):
        return self.P, self.q, self.G, self.h, None, None, None, None


@dataclass
class Solution:
    problem: Problem
    x: Optional[np.ndarray] = None
    found: bool = False


def solve_problem(problem: Problem, solver: str, **kwargs) -> Solution:
    if solver != "quadprog":
        raise ValueError(f"stub qpsolvers only restates 'quadprog', got '{solver}'")
    try:
        x = qp_gi.solve_qp(problem.P, problem.q, problem.G, problem.h)
    except qp_gi.NotPositiveDefinite as e:
        raise ProblemError(str(e))
    except qp_gi.Infeasible as e:
        warnings.warn(f"quadprog raised a ValueError: {e}")
        return Solution(problem)
    return Solution(problem, x=x, found=True)
