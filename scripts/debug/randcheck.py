import ctypes as C, sys
sys.path.insert(0, '.')
libc = C.CDLL("libc.so.6")
from speaker_recognition_amd import _lib
L = _lib.lib()
print("after loading the library: next rand() would be checked after HIP init")
print("device:", _lib.device_name())
vals = [libc.rand() for _ in range(3)]
print("rand() after HIP init:", vals, "(a fresh process gives 1804289383, 846930886, 1681692777)")
