"""The file a maintainer drops into the reference tree as ``models/M2Trans_network.py``.

``train.py:70`` resolves ``utils.import_module('models.{}_network'.format(args.model)).create_model(args)`` and
``test.py:16`` imports the class ``M2Trans`` from this module; both now reach the MI355X build.  Adjust the path below to
where this repository lives (here: two levels above ``integration/models/``).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from m2trans_amd.M2Trans_network import create_model, M2Trans  # noqa: E402,F401
