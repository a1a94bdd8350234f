#!/usr/bin/env python3
"""prints a tools/md_chain2.py result compactly"""
import json, sys
d = json.load(open(sys.argv[1]))
full = len(sys.argv) > 2
for p in d['pictures']:
    print(p['picture'], 'layer', p['temporal_layer'], 'identical', p['decisions_identical'], 'ms', p['kernel_ms'], 'ep_ms', p.get('ep_kernel_ms'), 'path', p['longest_path_ms'], 'units', p['units_on_path'], 'clk/unit', p['clocks_per_unit_on_path'])
    print('  stages', p['stage_clocks_per_unit_on_path'])
    print('  sub', p['sub_stage_clocks_per_unit_on_path'])
    if full:
        for k, v in p['by_depth_on_path'].items():
            print('  ', k, 'units', v['units'], 'share', v['share_of_path'], 'clk/unit', v['clocks_per_unit'], 'cands', v['candidates_per_unit'])
            print('      ', v['stages'])
            print('      sub', v['sub'])
