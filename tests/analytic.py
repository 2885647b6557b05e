"""Analytic fields and error norms of the reference operator test (test/utilities.jl), in numpy.

Data only + formulas: h / F / grad h / div F / curl F of utilities.jl:93-190 and
ErrorMeasures of utilities.jl:18-34.  Arrays are (n, K) == Julia (K, n)."""
import math

import numpy as np


class PlanarSetup:
    """TestSetup(Mesh, PlanarTest) -- utilities.jl:57-91."""

    def __init__(self, mesh, K):
        self.m, self.K = mesh, K
        self.Lx = float(round(mesh.xCell.max()))          # :71
        self.Ly = math.sqrt(3.0) / 2.0 * self.Lx          # :72
        self.nx, self.ny = np.cos(mesh.angleEdge), np.sin(mesh.angleEdge)

    def _tile(self, a):
        return np.repeat(a[:, None], self.K, axis=1)

    def h(self):                                           # :96-105
        m = self.m
        return self._tile(np.sin(2.0 * np.pi * m.xCell / self.Lx) * np.sin(2.0 * np.pi * m.yCell / self.Ly))

    def F_edge(self):                                      # :109-123, :159-173
        m = self.m
        Fx = np.sin(2.0 * np.pi * m.xEdge / self.Lx) * np.cos(2.0 * np.pi * m.yEdge / self.Ly)
        Fy = np.cos(2.0 * np.pi * m.xEdge / self.Lx) * np.sin(2.0 * np.pi * m.yEdge / self.Ly)
        return self._tile(self.nx * Fx + self.ny * Fy)

    def grad_h_edge(self):                                 # :125-135, :176-190
        m = self.m
        dx = 2.0 * np.pi / self.Lx * np.cos(2.0 * np.pi * m.xEdge / self.Lx) * np.sin(2.0 * np.pi * m.yEdge / self.Ly)
        dy = 2.0 * np.pi / self.Ly * np.sin(2.0 * np.pi * m.xEdge / self.Lx) * np.cos(2.0 * np.pi * m.yEdge / self.Ly)
        return self._tile(self.nx * dx + self.ny * dy)

    def div_F(self):                                       # :140-148
        m = self.m
        return self._tile(2.0 * np.pi * (1.0 / self.Lx + 1.0 / self.Ly) *
                          np.cos(2.0 * np.pi * m.xCell / self.Lx) * np.cos(2.0 * np.pi * m.yCell / self.Ly))

    def curl_F(self):                                      # :153-161
        m = self.m
        return self._tile(2.0 * np.pi * (-1.0 / self.Lx + 1.0 / self.Ly) *
                          np.sin(2.0 * np.pi * m.xVertex / self.Lx) * np.sin(2.0 * np.pi * m.yVertex / self.Ly))


def error_measures(numeric, analytic, area):
    """ErrorMeasures (utilities.jl:18-34): L_inf = |d|_inf/|a|_inf ; L_two = |d*area|_2/|a*area|_2."""
    d = analytic - numeric
    w = area[:, None]
    return (np.abs(d).max() / np.abs(analytic).max(),
            np.linalg.norm((d * w).ravel()) / np.linalg.norm((analytic * w).ravel()))


def areas(mesh):
    return {"cell": mesh.areaCell, "vertex": mesh.areaTriangle, "edge": mesh.dcEdge * mesh.dvEdge * 0.5}
