"""harmonic-power-flow on MI355X: the Newton-Raphson hot path of pweigmann/harmonic-power-flow's `hpf()` as
hand-written HIP (gfx950) behind the reference's own Python call shapes.  See DESIGN.md / INTEGRATION.md."""
from .settings import Settings
from .api import (AdmittanceSet, build_admittance_matrices, build_harmonic_jacobian, close_all, get_THD, handle_cache, harmonic_mismatch,
                  harmonic_state_vector, hpf, import_Norton_Equivalents, init_network, init_voltages, pf, solve,
                  update_harmonic_state_vec)
from .device import DeviceModel
from .ingest import export_Norton_Equivalents, read_Norton_file

__all__ = ["Settings", "AdmittanceSet", "DeviceModel", "build_admittance_matrices", "build_harmonic_jacobian", "close_all",
           "export_Norton_Equivalents", "get_THD", "handle_cache", "harmonic_mismatch", "harmonic_state_vector", "hpf", "import_Norton_Equivalents",
           "init_network", "init_voltages", "pf", "read_Norton_file", "solve", "update_harmonic_state_vec"]
