import os, sys, time
sys.path.insert(0, os.getcwd())
src = open("tools/rates.py").read()
head = src[:src.index("def timeit(")]
tail = src[src.index("# PCIe-inclusive: the host entry point sk_fused_pass"):]
exec(compile(head + "\ntable = synth.make_sheet(96, 8, dual=True, seed=4)\nctx.set_barcodes(table, 1)\n" + tail, "pcie", "exec"))
