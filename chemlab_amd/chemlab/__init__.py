"""py3 restatement of ChemLab's input stack for the in-scope configurations (SURVEY.md 8 a0/a15,
f-2): `@params` CLI, .gro/.top/.itp readers, topology -> interactions, reaction .cfg -> reactions.
Written from the file formats and the behaviour of /root/reference/src/chemlab/*.py (Python 2,
not importable under py3); pinned by tests/test_chemlab_inputs.py against fixtures in tests/golden/.
"""
