"""Handle create/destroy churn (GPU box): resident set size must level off.  Usage: python tools/handle_churn.py [cycles]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib, synth  # noqa: E402


def rss_kb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") // 1024


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    frames = synth.make_batch(7, 2)
    marks = []
    for i in range(n):
        h = _lib.Handle(max_frames=48)
        if i % 4 == 0:
            h.decode_batch(frames)
        h.close()
        if i in (10, n // 4, n // 2, 3 * n // 4, n - 1):
            marks.append((i, rss_kb()))
    print("cycle, RSS kB:", marks)
    grow = (marks[-1][1] - marks[2][1]) / max(1, marks[-1][0] - marks[2][0])
    print(f"growth over the second half: {grow:.2f} kB per create/destroy cycle")


if __name__ == "__main__":
    main()
