"""Fold superseded profile sets into one file:  archive_profiles.py <archive.md> <title> <file> [<file> ...]
Each section of the archive is a file verbatim (as profiles/archive_r01-r03.md was made); the files are removed from
the working tree (git rm them afterwards)."""
import os
import sys

archive, title, files = sys.argv[1], sys.argv[2], sorted(sys.argv[3:])
with open(archive, "w") as out:
    out.write(f"# {title}\n")
    out.write(f"{len(files)} files, in the order of their names; each section is the file verbatim.\n")
    for f in files:
        out.write(f"\n## {os.path.basename(f)}\n\n```\n")
        text = open(f).read()
        out.write(text if text.endswith("\n") else text + "\n")
        out.write("```\n")
for f in files:
    os.remove(f)
print(f"{len(files)} files -> {archive}")
