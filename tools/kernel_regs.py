"""Developer helper: VGPR / AGPR / spill counts of the kernels in a `--save-temps` gfx950 assembly file.
usage: python tools/kernel_regs.py <file.s> [name-substring]"""
import re
import sys

s = open(sys.argv[1]).read()
for blk in s.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print(name[:70], 'agpr', blk.split()[0], 'vgpr', re.search(r'\.vgpr_count:\s+(\d+)', blk).group(1), 'spill',
          re.search(r'\.vgpr_spill_count:\s+(\d+)', blk).group(1), 'lds',
          re.search(r'\.group_segment_fixed_size:\s+(\d+)', blk).group(1))
