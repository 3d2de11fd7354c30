python bench.py --no-cpu-baseline --no-extra-legs --steps 400 --warmup 5 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 6
for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -3 | tr '\n' ' '; echo; sleep 0.5; done
wait $BP
python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['repeats']['values'])"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3
