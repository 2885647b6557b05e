"""ESMF-like clock and alarms of the reference (src/infra/TimeManager.jl), host-only.

DateTime is `datetime.datetime`.  Julia's Dates periods become `Period(unit, value)`: Year / Month are calendar
periods (adding a month clamps the day to the month's length, like Dates), Day / Hour / Minute / Second are fixed and
interchangeable with `datetime.timedelta`, which is accepted wherever a period is.  `!` is dropped from names.
"""
from __future__ import annotations

import calendar
import datetime as _dt

__all__ = ["Period", "Year", "Month", "Day", "Hour", "Minute", "Second", "period_seconds", "Clock", "OneTimeAlarm",
           "PeriodicAlarm", "Alarm", "setCurrentTime", "changeTimeStep", "attachAlarm", "advance", "isRinging",
           "updateStatus", "rename", "stop", "reset", "mpas_create_clock", "TimeManagerError"]

_FIXED = {"Day": 86400, "Hour": 3600, "Minute": 60, "Second": 1}


class TimeManagerError(RuntimeError):
    """throw("...") of TimeManager.jl:178,183."""


class Period:
    """Dates.Year / Month / Day / Hour / Minute / Second."""

    __slots__ = ("unit", "value")

    def __init__(self, unit: str, value: int):
        if unit not in ("Year", "Month") and unit not in _FIXED:
            raise ValueError(f"unknown period unit {unit}")
        self.unit, self.value = unit, int(value)

    def __repr__(self):
        return f"{self.unit}({self.value})"

    def __eq__(self, other):
        if isinstance(other, Period):
            if self.unit == other.unit:
                return self.value == other.value
            if self.unit in _FIXED and other.unit in _FIXED:
                return self.total_seconds() == other.total_seconds()
            return False
        if isinstance(other, _dt.timedelta) and self.unit in _FIXED:
            return self.total_seconds() == other.total_seconds()
        return NotImplemented

    def __hash__(self):
        return hash((self.unit, self.value))

    def total_seconds(self) -> float:
        if self.unit not in _FIXED:
            raise TypeError(f"{self!r} has no fixed length")
        return float(self.value * _FIXED[self.unit])

    def timedelta(self) -> _dt.timedelta:
        return _dt.timedelta(seconds=self.total_seconds())

    def __radd__(self, t):                       # DateTime + Period
        if not isinstance(t, _dt.datetime):
            return NotImplemented
        if self.unit in _FIXED:
            return t + self.timedelta()
        months = self.value * (12 if self.unit == "Year" else 1)
        y, m0 = divmod(t.year * 12 + (t.month - 1) + months, 12)
        day = min(t.day, calendar.monthrange(y, m0 + 1)[1])          # Dates clamps to the last day of the month
        return t.replace(year=y, month=m0 + 1, day=day)

    def __rsub__(self, t):                       # DateTime - Period
        if not isinstance(t, _dt.datetime):
            return NotImplemented
        return t + Period(self.unit, -self.value)


def Year(n): return Period("Year", n)            # noqa: E704
def Month(n): return Period("Month", n)          # noqa: E704
def Day(n): return Period("Day", n)              # noqa: E704
def Hour(n): return Period("Hour", n)            # noqa: E704
def Minute(n): return Period("Minute", n)        # noqa: E704
def Second(n): return Period("Second", n)        # noqa: E704


def period_seconds(p) -> float:
    """Dates.value(Second(p)) as a float (mpas_ocean.jl:37)."""
    return float(p.total_seconds())


class Clock:                                      # TimeManager.jl:5-27
    def __init__(self, startTime: _dt.datetime, timeStep):
        self.startTime = startTime
        self.currTime = startTime
        self.prevTime = None                      # "at initialization there has been no prev. time"
        self.nextTime = startTime + timeStep
        self.timeStep = timeStep
        self.alarms = {}

    def __repr__(self):                           # Base.show, TimeManager.jl:67-73
        return (f"Simulation Clock with {len(self.alarms)} Alarms attached\n"
                f"├── Start Time   : {self.startTime}\n├── Current Time : {self.currTime}\n"
                f"├── Previous Time: {self.prevTime}\n├── Next Time    : {self.nextTime}\n"
                f"└── Timestep     : {self.timeStep}")


def setCurrentTime(clock: Clock, inCurrTime):     # :29-38 (an earlier time is only logged, nothing changes)
    if inCurrTime < clock.startTime:
        import logging
        logging.getLogger("moka_hip").error("Value of current time precedes start time")
        return
    clock.currTime = inCurrTime
    clock.prevTime = inCurrTime - clock.timeStep
    clock.nextTime = inCurrTime + clock.timeStep


def changeTimeStep(clock: Clock, timestep):       # :40-45
    clock.timeStep = timestep
    clock.nextTime = clock.currTime + timestep


def attachAlarm(clock: Clock, alarm):             # :47-50
    clock.alarms[alarm.name] = alarm


def advance(clock: Clock):                        # :52-60
    clock.prevTime = clock.currTime
    clock.currTime = clock.nextTime
    clock.nextTime = clock.currTime + clock.timeStep
    for a in clock.alarms.values():
        updateStatus(a, clock.currTime)


class OneTimeAlarm:                               # :83-95
    def __init__(self, name: str, alarmTime):
        self.name, self.ringing, self.stopped, self.ringTime = name, False, False, alarmTime


class PeriodicAlarm:                              # :98-120: first ring one interval after intervalStart
    def __init__(self, name: str, alarmInterval, intervalStart):
        self.name, self.ringing, self.stopped = name, False, False
        self.ringTime = intervalStart + alarmInterval
        self.ringInterval = alarmInterval
        self.ringTimePrev = None


def Alarm(name, a, b=None):                       # :123-125
    return OneTimeAlarm(name, a) if b is None else PeriodicAlarm(name, a, b)


def isRinging(alarm) -> bool:                     # :128-130
    return alarm.ringing


def updateStatus(alarm, currentTime):             # :132-134: rings only on equality
    if alarm.ringTime == currentTime:
        alarm.ringing = True


def rename(alarm, newName: str):                  # :136-138
    alarm.name = newName


def stop(alarm):                                  # :140-142
    alarm.ringing = False


def reset(alarm, inTime=None):                    # :145-174
    stop(alarm)
    if isinstance(alarm, OneTimeAlarm):
        if inTime is None:
            alarm.stopped = True
        else:
            alarm.ringTime = inTime
        return
    if inTime is None:
        alarm.ringTimePrev = alarm.ringTime
        alarm.ringTime = alarm.ringTimePrev + alarm.ringInterval
        return
    if inTime < alarm.ringTime:
        import logging
        logging.getLogger("moka_hip").error("input time less than the current ring time")
        return
    while alarm.ringTime <= inTime:
        alarm.ringTimePrev = alarm.ringTime
        alarm.ringTime = alarm.ringTimePrev + alarm.ringInterval


def mpas_create_clock(timeStep, startTime, stopTime=None, runDuration=None) -> Clock:      # :176-191
    if runDuration is not None:
        stop_time = startTime + runDuration
        if stopTime is not None and not (stopTime != stop_time):     # the reference's (inverted) consistency test, kept
            raise TimeManagerError("stopTime and runDuration are inconsistent")
    elif stopTime is None:
        raise TimeManagerError(" neither stopTime nor runDuration are specified")
    return Clock(startTime, timeStep)
