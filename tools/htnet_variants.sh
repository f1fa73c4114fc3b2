for v in 8420 12420 16420 8410 12410 8400 12400; do
  echo "FWD_VARIANT=$v $(HERMNET_FWD_VARIANT=$v python tools/htnet_bench.py 10 2>/dev/null | tail -1)"
done
for v in 12420 16420; do
  echo "FWD_VARIANT_L0=$v + FWD=12420 $(HERMNET_FWD_VARIANT=12420 HERMNET_FWD_VARIANT_L0=$v python tools/htnet_bench.py 10 2>/dev/null | tail -1)"
done
