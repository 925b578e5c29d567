#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite database or *_kernel_stats.csv / *_kernel_trace.csv) into the text
table kept under profiles/.  Usage: tools/rocprof_summary.py <results.db | dir> [--last-step | --last-full-step] > profiles/rNN_x.txt"""
import glob
import os
import sqlite3
import sys


def from_db(path, last_step):
    c = sqlite3.connect(path)
    where, args = '', ()
    if last_step:
        starts = [r[0] for r in c.execute(
            "select d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id "
            "where s.kernel_name like '%k_permute_fmap%' order by d.start")]
        if starts:
            where, args = 'where d.start>=?', (starts[-1],)
    if last_step == 'full':   # everything between the last two solves = one whole bench step (round 6: a solve no longer ends with
        #                           k_pose_finalize -- the window runs from the last-but-one k_pose_init to the last one)
        ends = [r[0] for r in c.execute(
            "select d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id "
            "where s.kernel_name like '%k_pose_finalize%' order by d.start")]
        inits = [r[0] for r in c.execute(
            "select d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id "
            "where s.kernel_name like '%k_pose_init%' order by d.start")]
        if len(ends) >= 2:
            where, args = 'where d.start>=? and d.end<=?', (ends[-2], ends[-1])
        elif len(inits) >= 2:
            where, args = 'where d.start>=? and d.start<?', (inits[-2], inits[-1])
    tot = c.execute(f"select sum(d.end-d.start) from rocpd_kernel_dispatch d {where}", args).fetchone()[0]
    rows = c.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        f"from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id {where} "
        "group by s.kernel_name order by 3 desc", args).fetchall()
    return tot, rows


def main():
    path = sys.argv[1]
    last = 'full' if '--last-full-step' in sys.argv else '--last-step' in sys.argv
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, '**', '*_results.db'), recursive=True)[0]
    tot, rows = from_db(path, last)
    print(f'# source: {os.path.basename(path)}   scope: {"last whole bench step" if last == "full" else "last bench step from the correlation build on" if last else "whole run"}')
    print(f'# total kernel time {tot / 1e6:.3f} ms')
    print(f'{"total_ms":>10} {"pct":>6} {"calls":>7} {"avg_us":>10} {"min_us":>10} {"max_us":>10}  kernel')
    for name, n, s, a, mn, mx in rows:
        print(f'{s / 1e6:10.3f} {100 * s / tot:6.2f} {n:7d} {a / 1e3:10.1f} {mn / 1e3:10.1f} {mx / 1e3:10.1f}  {name[:140]}')


if __name__ == '__main__':
    main()
