"""How fast does the box take 1.5 GB into a file?  One writer (write), N writers (pwrite at disjoint offsets), and a
memory-mapped file filled by N threads.  usage: write_probe.py [dir]"""
import mmap, os, sys, threading, time
d = sys.argv[1] if len(sys.argv) > 1 else '/tmp/e2e'
os.makedirs(d, exist_ok=True)
total = 1536 << 20
chunk = 4 << 20
buf = (b'0123456789abcdef' * (chunk // 16))
path = os.path.join(d, 'write_probe.bin')

def fresh():
    if os.path.exists(path):
        os.remove(path)

def one_writer():
    fresh()
    t = time.time()
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    for _ in range(total // chunk):
        os.write(fd, buf)
    os.close(fd)
    return time.time() - t

def n_pwriters(n):
    fresh()
    t = time.time()
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    def work(k):
        for i in range(k, total // chunk, n):
            os.pwrite(fd, buf, i * chunk)
    th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
    [x.start() for x in th]; [x.join() for x in th]
    os.close(fd)
    return time.time() - t

def n_mmap(n):
    fresh()
    t = time.time()
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    os.ftruncate(fd, total)
    mm = mmap.mmap(fd, total)
    mv = memoryview(mm)
    def work(k):
        for i in range(k, total // chunk, n):
            mv[i * chunk:(i + 1) * chunk] = buf
    th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
    [x.start() for x in th]; [x.join() for x in th]
    mv.release(); mm.close(); os.close(fd)
    return time.time() - t

for rep in range(2):
    print("write, one thread        : %.3f s = %.2f GB/s" % ((lambda s: (s, total / s / 1e9))(one_writer())), flush=True)
    for n in (2, 4, 8):
        s = n_pwriters(n); print("pwrite, %d threads        : %.3f s = %.2f GB/s" % (n, s, total / s / 1e9), flush=True)
    for n in (1, 4, 8):
        s = n_mmap(n); print("mmap + copy, %d threads   : %.3f s = %.2f GB/s" % (n, s, total / s / 1e9), flush=True)
fresh()
