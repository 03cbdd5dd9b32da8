mkdir -p gpurun_out/r4b
(echo "== offer boards, lingering shares (cap 64, every 32)"; python tools/adapt_bench.py --check
 echo "== cap 48 every 24"; LBVH_OFFER_CAP=48 LBVH_OFFER_EVERY=24 python tools/adapt_bench.py
 echo "== cap 32 every 16"; LBVH_OFFER_CAP=32 LBVH_OFFER_EVERY=16 python tools/adapt_bench.py
 echo "== cap 96 every 32"; LBVH_OFFER_CAP=96 python tools/adapt_bench.py
 echo "== cap 64 every 32 no linger"; LBVH_OFFER_LINGER=0 python tools/adapt_bench.py
 echo "== round 3 kernels"; LBVH_ADAPT=0 python tools/adapt_bench.py) > gpurun_out/r4b/adapt5.txt 2>&1
cat gpurun_out/r4b/adapt5.txt
