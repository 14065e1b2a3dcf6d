"""Dev helper: reads a rocprofv3 --kernel-trace CSV of a bench.py run and prints, for the steady-state frames, every launch of
one frame in start order with its start offset and duration averaged over the frames (a frame = the launches between two
stream_resolve<false>).  usage: python tools/frame_timeline.py gpurun_out/<dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n[n.find('::') + 2:] if '::' in n else n
    return n.split('(')[0][:40]
ends = [i for i, r in enumerate(rows) if 'stream_resolve<false>' in r['Kernel_Name']]
frames = []
for a, b in zip(ends[:-1], ends[1:]):
    fr = [r for r in rows[a + 1:b + 1] if 'rocclr' not in r['Kernel_Name'] and 'unpack' not in r['Kernel_Name'] and 'at::' not in r['Kernel_Name']]
    frames.append(fr)
sig = lambda fr: tuple(short(r['Kernel_Name']) for r in fr)
common = max(set(map(sig, frames)), key=lambda s: sum(1 for fr in frames if sig(fr) == s))
sel = [fr for fr in frames if sig(fr) == common][2:]
print('%d frames averaged' % len(sel))
for i, name in enumerate(common):
    st = sum(int(fr[i]['Start_Timestamp']) - int(fr[0]['Start_Timestamp']) for fr in sel) / len(sel) / 1e3
    du = sum(int(fr[i]['End_Timestamp']) - int(fr[i]['Start_Timestamp']) for fr in sel) / len(sel) / 1e3
    print('%-42s start %8.1f  dur %8.1f us' % (name, st, du))
