"""Constants of the hot path, same names and values as reference config.py:1-92 (the reference's
`TrainingArguments` blocks, config.py:94-177, are training configuration and out of scope)."""

# OpenAI's pretrained implementation (reference config.py:6-7)
CLIP_MODEL = 'openai/clip-vit-large-patch14-336'
CLIP_EMBED_DIM = 1024

# Geocells path (config.py:35-36)
GEOCELL_PATH = 'data/geocells_2203.csv'       # PIGEON
GEOCELL_PATH_YFCC = 'data/geocells_yfcc.csv'  # PIGEOTTO

# Models (config.py:58-68)
CURRENT_SAVE_PATH = 'saved_models/WorldCLIP_head_landmarks.model'
PRETRAINED_CLIP = 'saved_models/StreetviewCLIP.model'
CLIP_PRETRAINED_HEAD = 'saved_models/New_Base_smooth_avg_MT_Geo_SV.model'
PRETRAINED_CLIP_YFCC = 'saved_models/WorldCLIP.model'
CLIP_PRETRAINED_HEAD_YFCC = 'saved_models/WorldCLIP_head.model'
CLIP_PRETRAINED_HEAD_YFCC_LANDMARKS = 'saved_models/WorldCLIP_head_landmarks.model'

# Embedding (config.py:71)
EMBED_BATCH_SIZE_PER_GPU = 512

# Evaluation batch per device (config.py:97-98: per_device_eval_batch_size=256)
EVAL_BATCH_SIZE_PER_GPU = 256

# Cluster refinement model (config.py:76-88)
PROTO_PATH = 'data/data_prototypes_2203.csv'
DATASET_PATH = 'data/hf_SVCLIP_2203'
PROTO_MODEL_PATH = 'saved_models/refiner/proto.refiner'
PROTO_PATH_YFCC = 'data/data_prototypes_YFCC.csv'
DATASET_PATH_YFCC = 'data/hf_YFCC'
PROTO_MODEL_YFCC_PATH = 'saved_models/refiner/proto_YFCC.refiner'
PROTO_PATH_LANDMARKS = 'data/data_prototypes_landmarks.csv'
DATASET_PATH_LANDMARKS = 'data/hf_landmarks'
PROTO_MODEL_LANDMARKS_PATH = 'saved_models/refiner/proto_landmarks.refiner'

# CLIP image normalisation (transformers OPENAI_CLIP_MEAN / OPENAI_CLIP_STD, used by CLIPProcessor at
# reference models/clip_embedder.py:52 and dataset_creation/finetune/embed_dataset.py:20)
OPENAI_CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
OPENAI_CLIP_STD = [0.26862954, 0.26130258, 0.27577711]

# Geoguessr score decay (reference config.py:52) and haversine label smoothing (config.py:55)
DECAY_CONSTANT = 1492.7
LABEL_SMOOTHING_CONSTANT = 65  # (PIGEOTTO), 75 (PIGEON)
