import sys; sys.path.insert(0,'.')
from medicalseg_amd.device import get_device
dev = get_device()
n = 2 << 30
p = dev.malloc(n); q = dev.malloc(n)
for name, fn, bytes_ in (("memset 2GiB (write)", lambda: dev.memset(p, 0, n), n), ("d2d 2GiB (read+write)", lambda: dev.d2d(q, p, n), 2*n)):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(5): fn()
    ms = dev.timer_stop()/5
    print("%s: %.3f ms  %.2f TB/s" % (name, ms, bytes_/ms/1e9))
