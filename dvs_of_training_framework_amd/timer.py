"""Stage timers (reference utils/timer.py).

``FakeTimer`` serves the contract of utils/timer.py:19-49 with a null object.  ``EventTimer`` replaces
``SynchronizedWallClockTimer`` (utils/timer.py:52-132), which calls
``torch.cuda.synchronize()`` around every start/stop and so serialises the
stream: here start/stop only record HIP events on the current stream and the
elapsed time is read back when ``log``/``elapsed`` is called.
"""
import torch


class _Null:
    """Accepts any call and does nothing; ``elapsed`` answers 0."""

    def __getattr__(self, name):
        return (lambda *a, **k: 0) if name == 'elapsed' else (lambda *a, **k: None)


class FakeTimer:
    """Timers switched off: the contract of the reference's no-op class
    (utils/timer.py:19-49 -- ``timers(name).start() / .stop() / .reset() /
    .elapsed()``, ``timers.log(names, ...)``, ``memory_usage()``) served by
    one shared null object."""
    _null = _Null()

    def __init__(self):
        self.timers = {}

    def __call__(self, name):
        return self.timers.setdefault(name, self._null)

    @staticmethod
    def memory_usage():
        return ''

    def log(self, *args, **kwargs):
        pass


class EventTimer:
    class Timer:
        def __init__(self, name):
            self.name_ = name
            self.pairs = []
            self.open = None

        def start(self):
            assert self.open is None, 'timer has already been started'
            self.open = torch.cuda.Event(enable_timing=True)
            self.open.record()

        def stop(self):
            assert self.open is not None, 'timer is not started'
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            self.pairs.append((self.open, end))
            self.open = None

        def reset(self):
            self.pairs, self.open = [], None

        def elapsed(self, reset=True):
            """Seconds; waits only for the recorded events."""
            total = 0.0
            for a, b in self.pairs:
                b.synchronize()
                total += a.elapsed_time(b) / 1000.0
            if reset:
                self.pairs = []
            return total

    def __init__(self):
        self.timers = {}

    def __call__(self, name):
        if name not in self.timers:
            self.timers[name] = self.Timer(name)
        return self.timers[name]

    @staticmethod
    def memory_usage():
        gib = 1024 ** 3
        return (f' | mem allocated {torch.cuda.memory_allocated() / gib:.3f}'
                f' GiB | max {torch.cuda.max_memory_allocated() / gib:.3f}'
                ' GiB')

    def log(self, names, normalizer=1.0, reset=True, memory_breakdown=False):
        assert normalizer > 0.0
        string = 'time (ms)'
        for name in names:
            if name in self.timers:
                ms = self.timers[name].elapsed(reset=reset) * 1000.0
                string += f' | {name}: {ms / normalizer:.2f}'
        if memory_breakdown:
            string += self.memory_usage()
        if (not torch.distributed.is_initialized()
                or torch.distributed.get_rank() == 0):
            print(string, flush=True)
