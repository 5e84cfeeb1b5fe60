import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


# Slack counters of the draw-by-draw tests (chaotic trajectories: a chain may part from the oracle once rounding
# differences have been amplified past a decision threshold).  What every such test observed is printed at the end of
# the session and written to tests/slack_counters.json, so the margins are visible and not only their bounds.
SLACK = []


def record_slack(name, value, bound, of=None):
    SLACK.append({'test': name, 'value': int(value), 'bound': bound, 'of': of})


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not SLACK:
        return
    import json
    terminalreporter.write_line('slack counters (value observed; bound asserted; out of):')
    for s in SLACK:
        terminalreporter.write_line('  %-90s %3d   %-6s %s' % (s['test'][:90], s['value'], s['bound'], s['of'] if s['of'] is not None else ''))
    try:
        with open(os.path.join(ROOT, 'tests', 'slack_counters.json'), 'w') as f:
            json.dump(SLACK, f, indent=1)
    except OSError:
        pass
