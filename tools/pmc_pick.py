"""print, space-separated, those of the candidate counter names that `rocprofv3 -L` (saved to a file) offers.
usage: pmc_pick.py counters_list.txt NAME ..."""
import re, sys
text = open(sys.argv[1], errors="replace").read()
have = set(re.findall(r"[A-Za-z][A-Za-z0-9_]+", text))
print(" ".join(c for c in sys.argv[2:] if c in have))
