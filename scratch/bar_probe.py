"""Is hipMalloc memory writable from the CPU on this box (large BAR)?  Run in a subprocess: a failure is a segfault."""
import ctypes as C, subprocess, sys
if len(sys.argv) > 1:
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    flag = int(sys.argv[1])
    if flag == 0:
        rc = hip.hipMalloc(C.byref(p), 4096)
    else:
        hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        rc = hip.hipExtMallocWithFlags(C.byref(p), 4096, flag)
    attr = C.c_int(0)
    hip.hipDeviceGetAttribute(C.byref(attr), 87, 0)
    print("alloc rc", rc, hex(p.value or 0), flush=True)
    C.c_uint64.from_address(p.value).value = 0x1234567
    out = C.c_uint64(0)
    hip.hipMemcpy(C.byref(out), p, 8, 2)
    print("cpu store visible to hipMemcpy:", hex(out.value), flush=True)
    v = C.c_uint64(0x7654321)
    hip.hipMemcpy(p, C.byref(v), 8, 1)
    hip.hipDeviceSynchronize()
    print("cpu load sees hipMemcpy:", hex(C.c_uint64.from_address(p.value).value), flush=True)
else:
    for flag in (0, 1, 3):   # hipMalloc, hipDeviceMallocFinegrained = 0x1, hipDeviceMallocUncached = 0x3
        r = subprocess.run([sys.executable, __file__, str(flag)], capture_output=True, text=True)
        print("flag", flag, "rc", r.returncode, r.stdout.strip().replace("\n", " | "), r.stderr.strip()[-200:])
