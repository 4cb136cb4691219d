"""The US federal holiday calendar the citi_bike business engine stamps on every station
(``station.holiday = cur_datetime in holidays.US()``, maro/simulator/scenarios/citi_bike/business_engine.py:75, 382-394) —
restated natively, because the `holidays` package is a dependency this engine does not want (and cannot have offline).

``holidays.US()`` with no subdivision and ``observed=True`` (the form the reference constructs) holds the federal holidays and
their observed days:

    New Year's Day           January 1           (observed: Friday December 31 of the year before / Monday January 2)
    Martin Luther King Jr.   3rd Monday of January      (since 1986)
    Washington's Birthday    3rd Monday of February     (since 1971; February 22 before)
    Memorial Day             last Monday of May         (since 1971; May 30 before)
    Juneteenth               June 19             (since 2021; observed Friday / Monday)
    Independence Day         July 4              (observed Friday July 3 / Monday July 5)
    Labor Day                1st Monday of September
    Columbus Day             2nd Monday of October      (since 1971; October 12 before)
    Veterans Day             November 11         (observed Friday November 10 / Monday November 12)
    Thanksgiving             4th Thursday of November
    Christmas Day            December 25         (observed Friday December 24 / Monday December 26)

A holiday that falls on a Saturday is also observed on the Friday before, one on a Sunday on the Monday after; both the day
itself and the observed day are members.  tests/test_citi_bike_calendar.py pins every member from 2018-01 to 2021-12 (the span
of the reference's ny.201801 ... ny.202006 topologies and beyond).
"""
from datetime import date, timedelta


def _nth_weekday(year: int, month: int, weekday: int, n: int) -> date:
    """The n-th `weekday` (Monday = 0) of a month; n = -1: the last."""
    if n > 0:
        d = date(year, month, 1)
        return d + timedelta(days=(weekday - d.weekday()) % 7 + 7 * (n - 1))
    d = date(year + (month == 12), month % 12 + 1, 1) - timedelta(days=1)
    return d - timedelta(days=(d.weekday() - weekday) % 7)


def _with_observed(d: date):
    out = [d]
    if d.weekday() == 5:
        out.append(d - timedelta(days=1))
    elif d.weekday() == 6:
        out.append(d + timedelta(days=1))
    return out


def us_holidays(year: int) -> set:
    """Every date of `year` that is in ``holidays.US()`` (federal holidays + observed days; a New Year's Day that falls on a
    Saturday puts December 31 of the year BEFORE into that year's set)."""
    fixed = [date(year, 1, 1), date(year, 7, 4), date(year, 11, 11), date(year, 12, 25)]
    if year >= 2021:
        fixed.append(date(year, 6, 19))
    days = set()
    for d in fixed:
        days.update(x for x in _with_observed(d) if x.year == year)
    if date(year + 1, 1, 1).weekday() == 5:
        days.add(date(year, 12, 31))
    if year >= 1986:
        days.add(_nth_weekday(year, 1, 0, 3))
    days.add(_nth_weekday(year, 2, 0, 3) if year >= 1971 else date(year, 2, 22))
    days.add(_nth_weekday(year, 5, 0, -1) if year >= 1971 else date(year, 5, 30))
    days.add(_nth_weekday(year, 9, 0, 1))
    days.add(_nth_weekday(year, 10, 0, 2) if year >= 1971 else date(year, 10, 12))
    days.add(_nth_weekday(year, 11, 3, 4))
    return days


_CACHE = {}


def is_us_holiday(d) -> bool:
    """``d in holidays.US()`` for a date / datetime."""
    d = date(d.year, d.month, d.day)
    if d.year not in _CACHE:
        _CACHE[d.year] = us_holidays(d.year)
    return d in _CACHE[d.year]
