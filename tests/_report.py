"""Figures the at-size tests measure (RMS deviations, counts): printed where they are measured (visible with -s) AND collected
for the terminal summary, which pytest prints with -q too -- the driver's GPU test record then holds the per-configuration
error figures, not only pass / fail.  conftest.py prints the section and writes the JSON file."""
FIGURES = []


def note(line):
    FIGURES.append(str(line))
    print(line)
