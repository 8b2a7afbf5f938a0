"""Shared test helpers: import the hyphenated package dir, the oracle binding and the synthetic generator."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def plslam():
    return _load("plslam_amd", os.path.join(ROOT, "pl-slam_amd", "__init__.py"))


def synth():
    return _load("plslam_amd_synth", os.path.join(ROOT, "pl-slam_amd", "synth.py"))


def oracle():
    return _load("plo", os.path.join(ROOT, "oracle", "plo.py"))


EMU_LIB = os.path.join(ROOT, "tests", "hipemu", "_build", "libplslam_emu.so")
