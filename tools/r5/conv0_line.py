import json, sys
d = json.loads(sys.stdin.read()); h = d['roofline']['hbm_bound_kernels']
f, b = h['conv0_gn_gelu_fwd'], h['conv0_gn_gelu_bwd']
print(d['ms_per_step'], 'ms | conv0 device fwd', f.get('device_ms'), 'ms', f.get('device_frac_of_8TBps'), '| bwd', b.get('device_ms'), 'ms', b.get('device_frac_of_8TBps'),
      '| in-step events fwd', f['ms'], f['frac_of_8TBps'], 'bwd', b['ms'], b['frac_of_8TBps'], '| frontend', d['roofline']['frontend']['ms'], d['roofline']['frontend']['frac_of_8TBps'])
