"""MI355X-native time-stepping path of EnergyBalanceModel.jl (MIZ and classic models).

Host-side mirror of the reference's ``step!`` / ``integrate`` surface over hand-written HIP
kernels reached through the C ABI of ``include/ebm_hip.h``.  See DESIGN.md.
"""
from ._lib import EBMError, StaleFieldError, LIB_PATH, EXPORTS  # noqa: F401
from ._devices import visible_gpu_count, free_port  # noqa: F401
from .engine import Engine, cos2pit  # noqa: F401
from .infrastructure import (  # noqa: F401
    Collection, SpaceTime, Forcing, Solutions, default_parval, miz_paramset, classic_paramset,
    default_parameters, step_, integrate, reset_step_state, classic_time_index,
    MIZ_SOLVARS, CLASSIC_SOLVARS,
)
from .ensemble import (  # noqa: F401
    EnsembleRun, shard_columns, gather_columns, broadcast_inputs, hemispheric_mean,
)

Vec = "numpy.ndarray[float64]"  # the reference's Vec = Vector{Float64} (src/infrastructure.jl:13)
