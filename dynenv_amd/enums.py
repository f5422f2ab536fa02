"""Enumerations with the reference's names and values (DynEnv/cutils.py:10-64)."""
from enum import IntEnum


class _ArgEnum(IntEnum):
    def __str__(self):
        return self.name.lower()

    def __repr__(self):
        return str(self)

    @classmethod
    def argparse(cls, s):
        try:
            return cls[s.upper()]
        except KeyError:
            return s


class DynEnvType(_ArgEnum):
    ROBO_CUP = 0
    DRIVE = 1


class NoiseType(_ArgEnum):
    RANDOM = 0
    REALISTIC = 1


class ObservationType(_ArgEnum):
    FULL = 0
    PARTIAL = 1
    IMAGE = 2
