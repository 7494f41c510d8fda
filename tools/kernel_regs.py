"""Registers / LDS / scratch of the kernels in a -save-temps assembly file:  python tools/kernel_regs.py <file.s> [name filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    sh = re.search(r"(\w+?_body)", name)
    g = lambda k: re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body).group(1)
    dims = re.findall(r"(?:Dims|FullDims|CentDims)ILi(\d+)ELi(\d+)(?:ELi(\d+))?", name)
    print("%-22s %-5s %-24s vgpr %4s sgpr %4s lds %6s scratch %5s" % (sh.group(1) if sh else name[:22], "aux" if "ELi1EEE" in name else "", str(dims[:2]), g("next_free_vgpr"), g("next_free_sgpr"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
