import sys
sys.path.insert(0, '.')
from dlimgedit_amd import api
for n in (1, 2, 3, 4, 6, 8):
    pre_ms, post_ms = api.ext.bench_prepost(n, 60, 768)
    b = n * (256*256*4 + 1024*1024)
    print(f"n={n}: post {post_ms*1e3:.1f} us {b/post_ms/1e6:.0f} GB/s | pre {pre_ms*1e3:.1f} us")
