"""The JSON line of bench.py and the wall-clock budget of its optional legs.

The record must not be losable: the headline (metric, value, config, roofline, exchange.* at N > 1) is written as soon as
it exists, and every leg that finishes afterwards re-writes the WHOLE line, enriched — one complete JSON object per line,
the LAST complete line is the record.  A run that is killed inside an optional leg (a hang, the driver's limit) has
already left a valid line behind.  `Budget` decides, before a leg starts, whether the time left covers its estimate;
skipped legs are named in the line (`legs.skipped`), so a short record says why it is short.
"""
import json
import os
import time


class Budget:
    """Wall-clock budget of the whole run, counted from process start (`t0`: a time.monotonic() value)."""

    def __init__(self, total_s, t0=None):
        self.total_s = float(total_s)
        self.t0 = time.monotonic() if t0 is None else t0
        self.spent = {}          # leg -> seconds it took
        self.skipped = {}        # leg -> why

    def elapsed(self):
        return time.monotonic() - self.t0

    def remaining(self):
        return self.total_s - self.elapsed()

    def allows(self, leg, estimate_s):
        """True when `estimate_s` seconds still fit; otherwise the leg is recorded as skipped."""
        left = self.remaining()
        if left >= estimate_s:
            return True
        self.skipped[leg] = "skipped: needs ~%.0f s, %.0f s of the %.0f s budget left (--time-budget)" % (estimate_s, max(left, 0.0), self.total_s)
        return False

    def run(self, leg, estimate_s, fn, *args, **kw):
        """fn(*args) if the budget allows it, timed; -> its result, or None when skipped.  Exceptions propagate."""
        if not self.allows(leg, estimate_s):
            return None
        t = time.monotonic()
        try:
            return fn(*args, **kw)
        finally:
            self.spent[leg] = time.monotonic() - t

    def as_dict(self):
        return {"time_budget_s": self.total_s, "elapsed_s": self.elapsed(), "seconds": dict(self.spent), "skipped": dict(self.skipped)}


class Emitter:
    """Writes the record to `fd`, one complete line per call; the last line written is the record."""

    def __init__(self, fd, budget=None):
        self.fd = fd
        self.budget = budget
        self.lines = 0

    def emit(self, record, stage, final=False):
        rec = dict(record)
        self.lines += 1
        rec["record"] = {"line": self.lines, "stage": stage, "final": bool(final),
                         "note": "bench.py re-writes the whole line after every leg: the LAST complete line is the record"}
        if self.budget is not None:
            rec["legs"] = self.budget.as_dict()
        data = (json.dumps(rec) + "\n").encode()
        while data:                      # one write() may be short on a pipe
            n = os.write(self.fd, data)
            data = data[n:]
        return rec


def last_record(text):
    """The last COMPLETE JSON line of a bench.py stdout capture (a killed run may end in half a line); None if there is none."""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if not line.startswith("{"):
            continue
        try:
            return json.loads(line)
        except ValueError:
            continue
    return None
