"""How fast does a pageable NumPy array reach the device?  plain copy vs hipHostRegister + async copy (diagnostic)."""
import time, numpy as np, torch, ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]; hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
for mb in (134, 536):
    a = np.random.default_rng(0).integers(0, 60000, mb * 1024 * 1024 // 2, dtype=np.uint16)
    d = torch.empty(a.nbytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); hip.hipMemcpy(d.data_ptr(), a.ctypes.data, a.nbytes, 1); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(mb, "MB plain hipMemcpy", round((t1 - t0) * 1e3, 2), "ms", round(a.nbytes / (t1 - t0) / 1e9, 1), "GB/s")
    for rep in range(3):
        t0 = time.perf_counter(); rc = hip.hipHostRegister(a.ctypes.data, a.nbytes, 0); t1 = time.perf_counter()
        hip.hipMemcpy(d.data_ptr(), a.ctypes.data, a.nbytes, 1); torch.cuda.synchronize(); t2 = time.perf_counter()
        hip.hipHostUnregister(a.ctypes.data); t3 = time.perf_counter()
        print(mb, "MB register", rc, round((t1 - t0) * 1e3, 2), "copy", round((t2 - t1) * 1e3, 2), "unregister", round((t3 - t2) * 1e3, 2), "ms; total GB/s", round(a.nbytes / (t3 - t0) / 1e9, 1))
