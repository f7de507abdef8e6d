"""Host-side plugin runtime (Component / DataPool / Profiler / Car) against fixture G2 captured from the
reference's core/car.py + core/datapool.py, plus the documented corner cases."""
import pytest

from conftest import load_golden
from triton_racer_sim_amd.core import Car, Component, DataPool, Profiler


class A(Component):
    def __init__(self, trace):
        super().__init__(inputs=["b/out"], outputs=["a/out"])
        self.k, self.trace = 0, trace

    def step(self, *args):
        self.trace.append(["A", args[0]])
        self.k += 1
        if self.k > 4:
            raise KeyboardInterrupt
        return (self.k * 10,)

    def getName(self):
        return "A"


class B(Component):
    def __init__(self, trace):
        super().__init__(inputs=["a/out"], outputs=["b/out"])
        self.k, self.trace = 0, trace

    def step(self, *args):
        self.trace.append(["B", args[0]])
        self.k += 1
        return None if self.k == 2 else (self.k,)

    def onShutdown(self):
        self.trace.append(["B.onShutdown", None])

    def getName(self):
        return "B"


def test_car_trace_equals_reference():
    g2 = load_golden("car_trace.json")
    trace = []
    car = Car(loop_hz=1e9, verbose=False)
    car.addComponent(A(trace))
    car.addComponent(B(trace))
    car.start()
    assert trace == g2["trace"]
    assert car.pool.pool == g2["pool"]


def test_component_copies_port_lists():
    ins, outs = ["x"], ["y"]
    c = Component(inputs=ins, outputs=outs)
    ins.append("z")
    assert c.step_inputs == ["x"] and c.step_outputs == ["y"] and c.threaded is False
    assert c.getName() == "Generic Component" and c.step() is None


def test_datapool_semantics(capsys):
    pool = DataPool()
    c = Component(inputs=["i1", "i2"], outputs=["o1", "o2"])
    pool.add(c)
    assert pool.get_inputs_for(c) == (None, None)
    pool.set_value("i2", 5)
    assert pool.get_inputs_for(c) == (None, 5)
    pool.store_outputs_for(c, None)                       # None return stores nothing
    assert pool.get_value("o1") is None
    pool.store_outputs_for(c, (1, 2, 3))                  # extra values are ignored, positional
    assert (pool.get_value("o1"), pool.get_value("o2")) == (1, 2)
    with pytest.raises(Exception):
        pool.store_outputs_for(c, (7,))                   # too short -> prints the part, raises bare Exception
    assert "storing output 2" in capsys.readouterr().out
    assert pool.get_value("o1") == 7                      # the first value was stored before the failure


def test_car_rejects_non_components_and_runs_threads():
    car = Car(loop_hz=1000, verbose=False)
    with pytest.raises(AssertionError):
        car.addComponent(object())

    class T(Component):
        def __init__(self):
            super().__init__(outputs=["t/out"], threaded=True)
            self.ran = False

        def thread_step(self):
            self.ran = True

        def step(self, *a):
            return (1,)

    t = T()
    car.addComponent(t)
    car.start(max_ticks=3)
    assert car.ticks == 3 and t.ran and car.pool.get_value("t/out") == 1


def test_profiler_records_last_step_ms():
    p, c = Profiler(), Component()
    p.watch(c)
    p.stop_watch(c)
    assert list(p.profiles) == ["Generic Component"] and p.profiles["Generic Component"] >= 0.0
