"""Full sweep of tests/recipe_disagreement.py (CPU, container or any host with gcc): the oracle's CONTRACT recipe -- what the HIP kernels
reproduce bit for bit -- against its REFERENCE_FLOATS recipe -- what the reference's compiled objects reproduce bit for bit -- on the
plugin's parameters, Es/N0 30 ... 8 dB, 128 channels x 2 s each.  Writes profiles/r05/recipe_disagreement.{json,md}.
Usage: python profiles/recipe_disagreement.py [channels]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import recipe_disagreement as rd  # noqa: E402

if __name__ == "__main__":
    channels = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    res = rd.sweep(channels=channels)
    out = os.path.join(ROOT, "profiles", "r05")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "recipe_disagreement.json"), "w") as f:
        json.dump({str(k): v for k, v in res.items()}, f, indent=1)
    md = rd.markdown(res)
    with open(os.path.join(out, "recipe_disagreement.md"), "w") as f:
        f.write("# Oracle CONTRACT recipe vs REFERENCE_FLOATS recipe, plugin parameters, %d channels x 2 s per Es/N0 (\"after lock\" = the second second)\n\n" % channels)
        f.write(md + "\n")
    print(md)
