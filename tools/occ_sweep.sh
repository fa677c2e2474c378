for b in 0 1 2; do
  echo -n "HEAT_AMD_STREAM_BLOCKS=$b  "
  if [ $b = 0 ]; then python3 tools/run_config.py 3 20 3 stream | sed 's/.*| wall/| wall/'; else HEAT_AMD_STREAM_BLOCKS=$b python3 tools/run_config.py 3 20 3 stream | sed 's/.*| wall/| wall/'; fi
done
