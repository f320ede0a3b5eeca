"""Explicit replacement for the module globals of the reference (HG:578-593) and the kwargs defaults of
`pf` (HG:244) / `hpf` (HG:511).  Defaults equal the reference's."""
from dataclasses import dataclass, field
from typing import List


@dataclass
class Settings:
    BASE_POWER: float = 1000          # W        HG:578
    BASE_VOLTAGE: float = 400         # V        HG:579
    H_MAX: int = 51                   #          HG:581
    NET_FREQ: int = 50                # Hz       HG:583
    thresh_f: float = 1e-6            #          HG:244
    max_iter_f: int = 30
    thresh_h: float = 1e-4            #          HG:511
    max_iter_h: int = 50
    harmonics: List[int] = field(default=None)

    def __post_init__(self):
        if self.harmonics is None:
            self.harmonics = [h for h in range(1, self.H_MAX + 1, 2)]        # HG:584

    @property
    def HARMONICS(self):
        return self.harmonics

    @property
    def HARMONICS_FREQ(self):
        return [self.NET_FREQ * i for i in self.harmonics]                    # HG:585

    # p.u. system, HG:590-593 (same expressions, same doubles)
    @property
    def base_current(self):
        return self.BASE_POWER / self.BASE_VOLTAGE

    @property
    def base_admittance(self):
        return self.base_current / self.BASE_VOLTAGE

    @property
    def base_impedance(self):
        return 1 / self.base_admittance
