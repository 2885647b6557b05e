"""YAML configuration of the reference (src/infra/Config.jl): the Omega-style `omega:` tree split into namelist and
streams, with MPAS timestamps / time intervals parsed into datetimes and periods.  Host-only."""
from __future__ import annotations

import datetime as _dt
import logging
import os
import re

import yaml

from .timemanager import Day, Hour, Minute, Month, Second, Year

__all__ = ["yaml_config", "GlobalConfig", "ConfigRead", "ConfigGet", "ConfigAdd", "ConfigSet", "ConfigError",
           "DateTime_from_String", "parse_Datetimes"]


class ConfigError(RuntimeError):
    """error("...") of Config.jl:63,84,100."""


class yaml_config:                                # Config.jl:12-14
    def __init__(self, d: dict | None = None):
        self.dict = {} if d is None else d


class GlobalConfig:                               # Config.jl:24-27
    def __init__(self, namelist: yaml_config | None = None, streams: yaml_config | None = None):
        self.namelist = namelist if namelist is not None else yaml_config()
        self.streams = streams if streams is not None else yaml_config()


def ConfigGet(d: yaml_config, s: str):            # Config.jl:43-57: a sub-tree comes back wrapped, a leaf as is
    c = d.dict[s]                                 # KeyError, like the reference's dict lookup
    return type(d)(c) if isinstance(c, dict) else c


def ConfigAdd(d: yaml_config, s: str, val):       # Config.jl:61-68
    if s in d.dict:
        raise ConfigError(f"ConfigAdd: variable {s} already exists use ConfigSet instead")
    d.dict[s] = val


def ConfigSet(d: yaml_config, s: str, val):       # Config.jl:72-88 (a change of type is only warned about)
    if s not in d.dict:
        raise ConfigError(f"ConfigSet: Could not find variable {s}")
    if type(d.dict[s]) is not type(val):
        logging.getLogger("moka_hip").warning('ConfigSet: Changing typeof "%s", %s != %s', s, type(d.dict[s]).__name__,
                                              type(val).__name__)
    d.dict[s] = val


# YAML.jl reads `1.e25` as a float; PyYAML's YAML-1.1 resolver wants a signed exponent.  Core-schema floats instead.
class _Loader(yaml.SafeLoader):
    pass


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:\.[0-9]+|[0-9]+(?:\.[0-9]*)?)(?:[eE][-+]?[0-9]+)?$|^[-+]?\.(?:inf|Inf|INF)$|^\.(?:nan|NaN|NAN)$"),
    list("-+0123456789."))


def ConfigRead(filepath) -> GlobalConfig:         # Config.jl:98-119
    if not os.path.isfile(filepath):
        raise ConfigError("YAML configuration file does not exist")
    with open(filepath) as fh:
        config = yaml.load(fh, Loader=_Loader)    # noqa: S506 (SafeLoader subclass)
    streams = config["omega"].pop("streams")
    namelist = config.pop("omega")
    return GlobalConfig(yaml_config(parse_Datetimes(namelist)), yaml_config(parse_Datetimes(streams)))


def parse_Datetimes(d: dict) -> dict:             # Config.jl:121-138
    for key, value in d.items():
        if isinstance(value, dict):
            parse_Datetimes(value)
        elif isinstance(value, str) and timestamp_pat.search(value):
            d[key] = DateTime_from_String(value)
    return d


# Config.jl:142-151: [[[year-]month-]day][_]hh:mm:ss
timestamp_pat = re.compile(r"^(?:(?:(\d{1,4})-)?(?:(\d\d?)-)?(\d+))?_?(\d\d):(\d\d):(\d\d)$")
_PERIODS = (Year, Month, Day, Hour, Minute, Second)


def DateTime_from_String(string: str):            # Config.jl:166-224
    """A full non-zero date -> datetime; exactly one non-zero field -> that period; no date part (or a zero day
    count) -> datetime.time; anything else is warned about and returned unchanged."""
    mat = timestamp_pat.search(string)
    if mat is None:
        raise ConfigError("could not make sense of timestamp format")
    cap = mat.groups()
    if all(c is not None for c in cap):
        yr, mn, dy, h, m, s = (int(c) for c in cap)
        if mn != 0 and dy != 0:
            return _dt.datetime(yr, mn, dy, h, m, s)
    nums = [0 if c is None else int(c) for c in cap]
    if sum(1 for x in nums if x != 0) == 1:
        idx = next(i for i, x in enumerate(nums) if x != 0)
        return _PERIODS[idx](nums[idx])
    h, m, s = int(cap[3]), int(cap[4]), int(cap[5])
    if cap[0] is None and cap[1] is None and cap[2] is None:
        return _dt.time(h, m, s)
    if cap[0] is None and cap[1] is None and int(cap[2]) == 0:
        return _dt.time(h, m, s)
    logging.getLogger("moka_hip").warning(" Failed to parse %s ", string)
    return string
