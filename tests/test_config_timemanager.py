"""Config + Clock/Alarm layer, pinned by the reference's own tests:
test/infra/test_Config.jl (fixture tests/golden/config_test.yaml = its test.yaml) and test/infra/test_timeManager.jl."""
import datetime as dt
import os

import pytest

from moka_hip import api as mk
from moka_hip.config import (ConfigAdd, ConfigError, ConfigGet, ConfigRead, ConfigSet, DateTime_from_String, GlobalConfig,
                             yaml_config)
from moka_hip.timemanager import (Alarm, Clock, Day, Hour, Minute, Month, Second, TimeManagerError, Year, advance, attachAlarm,
                                  changeTimeStep, isRinging, mpas_create_clock, reset, setCurrentTime, stop)

HERE = os.path.dirname(__file__)


def test_config_read_matches_reference_test():
    """test/infra/test_Config.jl, line for line."""
    config = ConfigRead(os.path.join(HERE, "golden", "config_test.yaml"))
    hmix = ConfigGet(config.namelist, "hmix")
    intervals = ConfigGet(config.streams, "intervals")
    datetimes = ConfigGet(config.streams, "datetimes")
    assert isinstance(hmix, yaml_config)
    assert ConfigGet(hmix, "hmix_String") == "Restart_timestamp"
    assert ConfigGet(hmix, "hmix_Float") == 1.234567890
    assert ConfigGet(hmix, "hmix_None") == "none"
    assert ConfigGet(hmix, "hmix_On") is True
    assert ConfigGet(hmix, "hmix_Off") is False
    assert ConfigGet(hmix, "hmix_Exp") == 1.e25 and isinstance(ConfigGet(hmix, "hmix_Exp"), float)
    assert ConfigGet(intervals, "yearly_interval") == Year(1)
    assert ConfigGet(intervals, "monthly_interval") == Month(2)
    assert ConfigGet(intervals, "daily_interval") == Day(3)
    assert ConfigGet(intervals, "hourly_interval") == Hour(4)
    assert ConfigGet(intervals, "minutes_interval") == Minute(5)
    assert ConfigGet(intervals, "seconds_interval") == Second(6)
    assert ConfigGet(datetimes, "NO_HMS") == dt.datetime(1, 1, 1, 0, 0, 0)
    assert ConfigGet(datetimes, "NO_MS") == dt.datetime(1, 1, 1, 2, 0, 0)
    assert ConfigGet(datetimes, "NO_S") == dt.datetime(1, 1, 1, 2, 3, 0)
    assert ConfigGet(datetimes, "NO_H") == dt.datetime(1, 1, 1, 0, 3, 4)
    assert ConfigGet(datetimes, "NO_HM") == dt.datetime(1, 1, 1, 0, 0, 4)
    assert ConfigGet(datetimes, "NO_HS") == dt.datetime(1, 1, 1, 0, 3, 0)
    assert ConfigGet(datetimes, "ALL_HMS") == dt.datetime(1, 1, 1, 2, 3, 4)
    # "streams" was popped out of the namelist (Config.jl:108-110)
    assert "streams" not in config.namelist.dict and "omega" not in config.namelist.dict


def test_config_set_add_and_errors(tmp_path):
    c = yaml_config({"a": 1})
    ConfigAdd(c, "b", 2.0)
    with pytest.raises(ConfigError, match="already exists"):
        ConfigAdd(c, "b", 3.0)                                        # Config.jl:63
    ConfigSet(c, "a", 5)
    assert c.dict["a"] == 5
    ConfigSet(c, "a", "five")                                         # type change is only warned about (:77-81)
    with pytest.raises(ConfigError, match="Could not find"):
        ConfigSet(c, "zzz", 1)                                        # :84
    with pytest.raises(ConfigError, match="does not exist"):
        ConfigRead(tmp_path / "nope.yml")                             # :100
    assert isinstance(GlobalConfig().namelist, yaml_config)
    with pytest.raises(KeyError):
        ConfigGet(c, "missing")


def test_timestamp_forms():
    """Config.jl:166-224 branch by branch."""
    assert DateTime_from_String("2000-01-01_00:00:00") == dt.datetime(2000, 1, 1)
    assert DateTime_from_String("0000-00-00_10:00:00") == Hour(10)
    assert DateTime_from_String("12:34:56") == dt.time(12, 34, 56)            # no date part -> Time
    assert DateTime_from_String("0_01:02:03") == dt.time(1, 2, 3)             # zero day count -> Time
    assert DateTime_from_String("5_00:00:00") == Day(5)
    assert DateTime_from_String("0000-00-00_10:30:00") == "0000-00-00_10:30:00"   # two non-zero fields: warned, unchanged


def test_clock_and_alarms_match_reference_test():
    """test/infra/test_timeManager.jl: two simulated years in 20-minute steps, every alarm rings when it must."""
    time0 = dt.datetime(2000, 1, 1)
    clock = Clock(time0, Hour(1))
    assert clock.currTime == time0 and clock.timeStep == Hour(1) and clock.prevTime is None
    t_mar, t_aug, t_ny = dt.datetime(2020, 3, 1), dt.datetime(2019, 8, 24), dt.datetime(2020, 1, 1)
    a_mar, a_aug, a_ny = Alarm("2020-03-01", t_mar), Alarm("2019-08-24", t_aug), Alarm("New Year 2020", t_ny)
    periodic = {"20min": Alarm("Every 20 minutes", Minute(20), time0), "1h": Alarm("Every hour", Hour(1), time0),
                "6h": Alarm("Every 6 hours", Hour(6), time0), "day": Alarm("Every day", Day(1), time0),
                "month": Alarm("Every month", Month(1), time0), "year": Alarm("Every year", Year(1), time0)}
    for a in (a_mar, a_aug, a_ny, *periodic.values()):
        attachAlarm(clock, a)
    changeTimeStep(clock, Minute(20))
    assert clock.timeStep == Minute(20)
    cur = dt.datetime(2019, 1, 1)
    setCurrentTime(clock, cur)
    assert (clock.currTime, clock.prevTime, clock.nextTime) == (cur, dt.datetime(2018, 12, 31, 23, 40), dt.datetime(2019, 1, 1, 0, 20))
    for a in periodic.values():
        reset(a, cur)
    stop_time = dt.datetime(2021, 1, 1)
    seen = {k: 0 for k in list(periodic) + ["mar", "aug", "ny"]}
    while clock.currTime <= stop_time:
        advance(clock)
        t = clock.currTime
        for tag, tt, a in (("mar", t_mar, a_mar), ("aug", t_aug, a_aug), ("ny", t_ny, a_ny)):
            if t == tt:
                assert isRinging(a)
                stop(a)
                seen[tag] += 1
        checks = {"20min": t.minute % 20 == 0 and t.second == 0, "1h": t.minute == 0 and t.second == 0,
                  "6h": t.hour % 6 == 0 and t.minute == 0, "day": t.hour == 0 and t.minute == 0,
                  "month": t.day == 1 and t.hour == 0 and t.minute == 0,
                  "year": t.month == 1 and t.day == 1 and t.hour == 0 and t.minute == 0}
        for k, due in checks.items():
            if due:
                assert isRinging(periodic[k]), (k, t)
                reset(periodic[k])               # periodic alarms ring on equality only: re-arm for the next interval
                seen[k] += 1
            else:
                assert not isRinging(periodic[k]), (k, t)
    assert seen["mar"] == seen["aug"] == seen["ny"] == 1
    assert seen["year"] == 2 and seen["month"] == 24 and seen["day"] == 731 and seen["1h"] == 731 * 24


def test_calendar_periods_and_create_clock():
    assert dt.datetime(2020, 1, 31) + Month(1) == dt.datetime(2020, 2, 29)     # Dates clamps to the month's last day
    assert dt.datetime(2020, 2, 29) + Year(1) == dt.datetime(2021, 2, 28)
    assert dt.datetime(2020, 3, 1) - Day(1) == dt.datetime(2020, 2, 29)
    assert Hour(1) == Minute(60) == dt.timedelta(hours=1) and Hour(1) != Month(1)
    with pytest.raises(TimeManagerError):
        mpas_create_clock(Hour(1), dt.datetime(1, 1, 1))                       # TimeManager.jl:183
    c = mpas_create_clock(Hour(1), dt.datetime(1, 1, 1), runDuration=Day(2))
    assert c.nextTime == dt.datetime(1, 1, 1, 1)


def test_ocn_setup_clock_from_global_config():
    nl = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_stop_time": "none",
                              "config_run_duration": Hour(10), "config_restart_timestamp_name": "Restart_timestamp",
                              "config_do_restart": False},
          "time_integration": {"config_dt": Minute(5), "config_number_of_time_levels": 2}}
    st = {"output": {"reference_time": dt.datetime(1, 1, 1), "output_interval": Hour(1), "filename_template": "o.nc"}}
    clock = mk.ocn_setup_clock(GlobalConfig(yaml_config(nl), yaml_config(st)))
    assert clock.alarms["simulation_end"].ringTime == dt.datetime(1, 1, 1, 10)          # init.jl:98
    assert clock.alarms["outputAlarm"].ringTime == dt.datetime(1, 1, 1, 1)              # first ring one interval in
    nl["time_management"]["config_run_duration"] = "none"
    with pytest.raises(mk.MokaError, match="Neither"):
        mk.ocn_setup_clock(GlobalConfig(yaml_config(nl), yaml_config(st)))              # init.jl:94
    nl["time_management"]["config_stop_time"] = dt.datetime(1, 1, 2)
    assert mk.ocn_setup_clock(GlobalConfig(yaml_config(nl), yaml_config(st))).alarms["simulation_end"].ringTime == dt.datetime(1, 1, 2)
