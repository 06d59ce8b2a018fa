"""levelsetmethods.jl_amd — MI355X-native grid-update hot path of LevelSetMethods.jl.

Hand-written HIP kernels for gfx950 (csrc/) behind the C ABI of include/lsm.h, plus the host-side
mirror of the reference's LevelSetEquation / integrate! / MeshField interface (api.py).

The directory name contains a dot, so import it through the repo-root shim: ``import lsm_amd``.
"""
from . import _lib
from ._lib import LsmError, build
from .api import (AdvectionTerm, BoundaryCondition, CartesianGrid, CurvatureTerm, EikonalReinitializationTerm,
                  ExtrapolationBC, ForwardEuler, LazyMeshField, LevelSetEquation, LevelSetTerm, LinearExtrapolationBC, MeshField,
                  NarrowBandMeshField, NeumannBC, NormalMotionTerm, PeriodicBC, RK2, RK3, RigidRotation, ROCMeshField,
                  ROCNarrowBandMeshField, SeparableCoefficient,
                  SymmetryBC, TimeIntegrator, Upwind, WENO5, current_state, current_time, extend_along_normals_, integrate_, reinitialize_,
                  perimeter, volume, InterpolatedField, NewtonSDF, hausdorff_distance, SideField, curvature, curvature_field, gradient, gradient_field, normal, normal_field,
                  vortex_deformation, show, LocalGroup, nodeindices, cellindices, getnode, getcell, active_nodeindices, active_cellindices,
                  update_band_)

__all__ = [
    "AdvectionTerm", "BoundaryCondition", "CartesianGrid", "CurvatureTerm", "EikonalReinitializationTerm",
    "ExtrapolationBC", "ForwardEuler", "LazyMeshField", "LevelSetEquation", "LevelSetTerm", "LinearExtrapolationBC", "MeshField",
    "NarrowBandMeshField", "ROCNarrowBandMeshField", "NeumannBC", "NormalMotionTerm", "PeriodicBC", "RK2", "RK3",
    "RigidRotation", "ROCMeshField",
    "SeparableCoefficient", "SymmetryBC", "TimeIntegrator", "Upwind", "WENO5", "current_state", "current_time",
    "integrate_", "vortex_deformation", "volume", "perimeter", "extend_along_normals_", "reinitialize_", "LsmError", "build",
    "InterpolatedField", "NewtonSDF", "hausdorff_distance", "SideField", "curvature", "curvature_field", "gradient", "gradient_field", "normal", "normal_field", "show", "LocalGroup",
    "nodeindices", "cellindices", "getnode", "getcell", "active_nodeindices", "active_cellindices", "update_band_",
]
