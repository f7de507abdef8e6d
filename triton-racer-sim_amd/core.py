"""Host-side plugin runtime with the reference's contract: ``Component``, ``DataPool``, ``Profiler``, ``Car``.

Behavioural contract (each clause checked against golden fixture G2, ``tests/golden/car_trace.json``):

* ``Component(inputs, outputs, threaded)`` keeps private copies of the two name lists as
  ``step_inputs`` / ``step_outputs`` (reference ``components/component.py:5-9``) and offers the
  lifecycle hooks ``onStart / step / thread_step / onShutdown / getName`` (``:11-28``).
* ``DataPool`` is a name→value blackboard: every declared name starts as ``None``
  (``core/datapool.py:7-12``); inputs are handed over as a tuple in declared order (``:14-17``);
  outputs are stored positionally and a ``None`` return stores nothing (``:19-25``); a storage
  failure prints the component name and raises a bare ``Exception`` (``:26-28``).
* ``Car`` ticks its parts sequentially at ``loop_hz`` (``core/car.py:43-65``): a part sees the
  same-tick outputs of earlier parts and the previous-tick outputs of later ones; the second
  overrun prints the per-part profile (``:57-62``); ``KeyboardInterrupt`` ends the loop and
  ``stop()`` always runs ``onShutdown`` of every part (``:67-70,79-82``).

If the reference package is importable (a real Triton-Racer-Sim checkout), ``Component`` IS the
reference's class, so parts defined here pass ``Car.addComponent``'s ``issubclass`` assert
(``core/car.py:17``) and drop into ``car_templates/manage.py`` unchanged.
"""
import threading
import time

try:  # inside a Triton-Racer-Sim checkout: subclass the real ABC
    from TritonRacerSim.components.component import Component as _RefComponent
except Exception:  # standalone
    _RefComponent = None


if _RefComponent is not None:
    Component = _RefComponent
else:
    class Component:
        """A named-port part of the car."""

        def __init__(self, inputs=(), outputs=(), threaded=False):
            self.step_inputs = list(inputs)
            self.step_outputs = list(outputs)
            self.threaded = threaded

        def onStart(self):
            """Runs once, before the first tick."""

        def step(self, *args):
            """One tick on the main loop; returns a tuple matching ``step_outputs`` or ``None``."""

        def thread_step(self):
            """Body of the part's own daemon thread (only when ``threaded``)."""

        def onShutdown(self):
            """Runs once, when the car stops."""

        def getName(self):
            return "Generic Component"


class DataPool:
    def __init__(self):
        self.pool = {}

    def add(self, component):
        for name in list(component.step_inputs) + list(component.step_outputs):
            self.pool[name] = None

    def get_inputs_for(self, component):
        return tuple(self.pool[name] for name in component.step_inputs)

    def store_outputs_for(self, component, output_values=None):
        if output_values is None:
            return
        slot = 0
        try:
            for slot, name in enumerate(component.step_outputs):
                self.pool[name] = output_values[slot]
        except Exception:
            print(f"Datapoll: error associated with {component.getName()} on storing output {slot + 1}")
            raise Exception()

    def get_value(self, name):
        return self.pool[name]

    def set_value(self, name, value):
        self.pool[name] = value


class Profiler:
    """Wall-clock milliseconds of each part's most recent ``step`` (reference ``core/profiler.py:9-18``)."""

    def __init__(self):
        self.profiles = {}
        self._t0 = 0.0

    def watch(self, component):
        self._t0 = time.time()

    def stop_watch(self, component):
        self.profiles[component.getName()] = (time.time() - self._t0) * 1000.0

    def dump(self):
        for name, ms in self.profiles.items():
            print(f"{name}: {ms} ms")


class Car:
    def __init__(self, loop_hz=30, verbose=True):
        self.pool = DataPool()
        self.components = []
        self.component_threads = []
        self.loop_hz = loop_hz
        self.profiler = Profiler()
        self.verbose = verbose
        self.ticks = 0

    def _say(self, msg):
        if self.verbose:
            print(msg)

    def addComponent(self, component):
        assert issubclass(type(component), Component)
        self.components.append(component)
        self.pool.add(component)
        self._say(f"Added component: {component.getName()}")
        if component.threaded:
            self.component_threads.append(threading.Thread(target=component.thread_step, args=(), daemon=True))

    def tick(self):
        """One pass over all parts — the body of the reference's hot loop (``core/car.py:45-53``)."""
        for part in self.components:
            args = self.pool.get_inputs_for(part)
            self.profiler.watch(part)
            produced = part.step(*args)
            self.profiler.stop_watch(part)
            self.pool.store_outputs_for(part, produced)
        self.ticks += 1

    def start(self, max_ticks=None):
        """Run until ``KeyboardInterrupt`` (or ``max_ticks``, an addition for unattended runs)."""
        for part in self.components:
            part.onStart()
        for t in self.component_threads:
            t.start()
        period = 1.0 / self.loop_hz
        overran_before = False
        try:
            while max_ticks is None or self.ticks < max_ticks:
                t0 = time.time()
                self.tick()
                spent = time.time() - t0
                if spent > period:
                    if overran_before:
                        print(f"Loop frequency compromised! Actual time: {spent * 1000} ms")
                        print("[Part Performances]")
                        self.profiler.dump()
                    overran_before = True
                else:
                    time.sleep(period - spent)
        except KeyboardInterrupt:
            pass
        finally:
            self.stop()

    def stop(self):
        self._say("[Stopping car]")
        for part in self.components:
            part.onShutdown()
