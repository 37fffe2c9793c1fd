"""Scratch probe: does hipExtStreamCreateWithCUMask work on this box?"""
import ctypes
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
s = ctypes.c_void_p()
mask = (ctypes.c_uint32 * 8)(*([0x0F0F0F0F] * 8))
rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(8), mask)
print("hipExtStreamCreateWithCUMask rc =", rc, "stream", s.value)
hip.hipGetErrorString.restype = ctypes.c_char_p
print(hip.hipGetErrorString(rc))
class Prop(ctypes.Structure): _fields_ = [("raw", ctypes.c_char * 4096)]
n = ctypes.c_int(); hip.hipDeviceGetAttribute(ctypes.byref(n), ctypes.c_int(63), ctypes.c_int(0)); print("attr63", n.value)
out = (ctypes.c_uint32 * 8)()
if rc == 0:
    rc2 = hip.hipExtStreamGetCUMask(s, ctypes.c_uint32(8), out)
    print("get mask rc", rc2, [hex(x) for x in out])
