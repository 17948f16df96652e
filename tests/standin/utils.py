"""Stand-in for the reference's ``node_classification_clean/utils.py`` next to ``time_model_standin.py``: the real one imports
torch_geometric / ogb (absent in the build container); a script run through ``python -m kagnn_amd.run_reference`` loads ITS OWN
directory's ``utils`` -- here this file -- exactly as ``python script.py`` would."""
dataset_layers = {"standin": 2}          # the table the timing script indexes (reference utils.py:17)
