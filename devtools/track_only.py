"""dev helper (GPU box): bench.py's value_track section alone.   python devtools/track_only.py [n_az [n_scans]]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

if __name__ == "__main__":
    n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 22
    out, _, _ = bench.track_section(n_az, n_scans, 1)
    print(json.dumps(out))
