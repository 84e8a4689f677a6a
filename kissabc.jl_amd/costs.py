"""DeviceCost: the `cost` argument of ApproxKernelizedPosterior / ApproxPosterior /
smc on the device path.  The reference takes an arbitrary Julia closure
(src/types.jl:124,137; src/smc.jl:94); a gfx950 kernel cannot call one, so a
cost is an id + parameter/data arrays whose formula lives in
include/kabc_costs.h (evaluated by the HIP kernels)."""
import ctypes as C

import numpy as np

from . import _cdefs as cd


class DeviceCost:
    def __init__(self, cost_id, params=(), data=(), name=None, dim=None):
        self.id = int(cost_id)
        self.params = np.ascontiguousarray(np.asarray(params, dtype=np.float64).ravel())
        self.data = np.ascontiguousarray(np.asarray(data, dtype=np.float64).ravel())
        self.name = name or f"cost{cost_id}"
        self.dim = dim

    def to_c(self):
        c = cd.Cost()
        c.id = self.id
        c.nparams = self.params.size
        c.params = self.params.ctypes.data_as(cd.c_double_p) if self.params.size else None
        c.ndata = self.data.size
        c.data = self.data.ctypes.data_as(cd.c_double_p) if self.data.size else None
        return c

    def __repr__(self):
        return f"DeviceCost({self.name})"


def GaussDist(center):
    """‖x − c‖₂"""
    return DeviceCost(cd.COST_GAUSS_DIST, params=center, name="gauss_dist")


def Rosenbrock():
    """sqrt(Σ 100(x[k+1]−x[k]²)² + (1−x[k])²) (configs C3, C5)"""
    return DeviceCost(cd.COST_ROSENBROCK, name="rosenbrock")


def HierGaussSim(ybar_obs):
    """θ = (m, s, z₁..z_G): ȳ_g = m + s z_g + randn/√8, cost = RMS(ȳ − ȳ_obs) (config C4)"""
    return DeviceCost(cd.COST_HIER_GAUSS_SIM, data=ybar_obs, name="hier_gauss_sim")


def NormalMeanStdSim(n, mean_obs, std_obs):
    """README.md:43-49: simulate n draws N(μ,σ); hypot(mean−mean_obs, 50(std−std_obs))"""
    return DeviceCost(cd.COST_NORMAL_MEANSTD_SIM, params=[n, mean_obs, std_obs],
                      name="normal_meanstd_sim")


def DiracSq(target=1.5):
    """test/runtests.jl:79-80: |μ²+1 − target|"""
    return DeviceCost(cd.COST_DIRAC_SQ, params=[target], name="dirac_sq")


def AbsDiff(target):
    """test/runtests.jl:178: |x − target|"""
    return DeviceCost(cd.COST_ABS_DIFF, params=[target], name="abs_diff")


def NormShell(target):
    """test/runtests.jl:186: |‖x‖₂ − target|"""
    return DeviceCost(cd.COST_NORM_SHELL, params=[target], name="norm_shell")


def NoisyQuadDU(target=5.5):
    """test/runtests.jl:108-109: |(n²+du)(n+0.01·randn) − target|"""
    return DeviceCost(cd.COST_NOISY_QUAD_DU, params=[target], name="noisy_quad_du")


def Mixture(target=0.0):
    """test/runtests.jl:145-146: |μ + rand((0.1·randn, randn)) − target|"""
    return DeviceCost(cd.COST_MIXTURE, params=[target], name="mixture")


def NoisyBanana(p_inf=0.0):
    """test/runtests.jl:242,248: 50(x+0.01z₁−y²)² + (y−1+0.01z₂)², +Inf with prob p_inf"""
    return DeviceCost(cd.COST_NOISY_BANANA, params=[p_inf], name="noisy_banana")


def WienerRms(tdata):
    """test/runtests.jl:116-126: mean |sqrt(μ²t²+σ²t)·(0.95+0.1·rand) − tdata_t|"""
    return DeviceCost(cd.COST_WIENER_RMS, data=tdata, name="wiener_rms")
