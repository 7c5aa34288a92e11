"""debug driver: tests/parity.check_plate_of_scans on the device with blocking launches"""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from genjax_amd import engine
jit = os.environ.get("DBG_JIT", "1") == "1"
engine.JIT_MIN_PARTICLES = 1024 if jit else 1 << 40
engine.JIT_MIN_WORK = 1024 if jit else 1 << 40
from tests import parity
n, no, T = int(os.environ.get("DBG_N", 5000)), int(os.environ.get("DBG_NO", 3)), int(os.environ.get("DBG_T", 40))
print(parity.check_plate_of_scans(n=n, no=no, T=T), flush=True)
