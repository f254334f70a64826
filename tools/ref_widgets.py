"""Headless stand-in for ipywidgets / IPython.display (THIS CONTAINER ONLY) so that the reference's
notebook-3 widget function launch_deform_viewer_fixed_camera (utils/deformation_estimation.py:15) can be
driven programmatically to capture golden vectors of its numeric closures."""
import sys
import types


class _Widget:
    def __init__(self, *children, **kw):
        self.children = children
        self._observers = []
        self._value = kw.get("value")
        self.options = kw.get("options")
        if self._value is None and self.options:
            self._value = list(self.options)[0]
        self.description = kw.get("description", "")

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, v):
        old = self._value
        self._value = v
        if v != old:
            for fn in list(self._observers):
                fn({"new": v, "old": old, "name": "value", "owner": self})

    def set_silently(self, v):
        self._value = v

    def observe(self, fn, names=None):
        self._observers.append(fn)


class _Button(_Widget):
    def __init__(self, **kw):
        super().__init__(**kw)
        self._clicks = []

    def on_click(self, fn):
        self._clicks.append(fn)

    def click(self):
        for fn in self._clicks:
            fn(self)


class _Output(_Widget):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


CREATED = {"buttons": [], "sliders": {}}


def _mk(cls, registry=None):
    def make(*a, **k):
        w = cls(*a, **k)
        if registry == "buttons":
            CREATED["buttons"].append(w)
        elif registry == "sliders":
            CREATED["sliders"][k.get("description", "")] = w
        return w
    return make


def install():
    CREATED["buttons"].clear(); CREATED["sliders"].clear()
    m = types.ModuleType("ipywidgets")
    m.FloatSlider = _mk(_Widget, "sliders"); m.IntSlider = _mk(_Widget, "sliders"); m.Dropdown = _mk(_Widget, "sliders")
    m.Button = _mk(_Button, "buttons"); m.Output = _mk(_Output)
    m.VBox = _mk(_Widget); m.HBox = _mk(_Widget); m.Layout = lambda **k: None
    m.Checkbox = _mk(_Widget, "sliders"); m.ToggleButtons = _mk(_Widget, "sliders"); m.Label = _mk(_Widget); m.HTML = _mk(_Widget)
    sys.modules["ipywidgets"] = m
    ip = types.ModuleType("IPython"); ip.get_ipython = lambda: None; ip.version_info = (8, 12, 3)
    disp = types.ModuleType("IPython.display"); disp.display = lambda *a, **k: None; disp.clear_output = lambda *a, **k: None
    ip.display = disp
    sys.modules["IPython"] = ip; sys.modules["IPython.display"] = disp
    return m


def closure_of(fn, name):
    """the inner function `name` captured in fn's closure cells"""
    for cell in fn.__closure__ or ():
        try:
            v = cell.cell_contents
        except ValueError:
            continue
        if callable(v) and getattr(v, "__name__", "") == name:
            return v
    raise KeyError(name)
