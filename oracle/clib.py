"""ORACLE (test infrastructure): ctypes loader of oracle/liboracle.so (gcc build of oracle/*.c)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    if not os.path.exists(_PATH):
        build()
    return C.CDLL(_PATH)


lib = load()
