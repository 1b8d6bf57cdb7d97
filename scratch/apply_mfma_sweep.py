p='quadruped-robot_amd/csrc/qr_mpc_kernel.hip'; s=open(p).read()
start = s.index("    // ---------------- phase 3: symmetric block sweep in registers,  A <- -H^-1 ----------------")
end = s.index("    QR_TS(3);")
old = s[start:end]
new = open('scratch/mfma_sweep_block.txt').read()
s = s[:start] + new + old + "#endif\n" + s[end:]
open(p,'w').write(s)
