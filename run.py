"""Command-line entry, same surface as the reference's run.py (:21-93): the mode is the first positional
argument (`pretrain | finetune | embed | evaluate`), then the model name/path.

Only the inference hot path is implemented here: `embed` (multi-GPU CLIP embedding of a dataset,
preprocessing/embed.py) and `evaluate` (SuperGuessr + ProtoRefiner, evaluation/evaluate.py).  `pretrain` and
`finetune` are training and out of scope -> NotImplementedError.

Because the reference's weights / geocells / data are not public (reference README.md:11) every input can be
replaced by seeded synthetic fixtures:  `--synthetic N` embeds / evaluates N synthetic panoramas.

  python run.py embed saved_models/StreetviewCLIP.model -l data/hf_dataset          # weights from a checkpoint
  python run.py embed random --synthetic 64                                        # plumbing check, random ViT
  python run.py evaluate saved_models/head.model -l data/hf_eval -b saved_models/StreetviewCLIP.model
  torchrun --nproc-per-node 8 --master-addr 127.0.0.1 run.py embed random --synthetic 4096
"""
import argparse
import logging
import os

import torch

logger = logging.getLogger('run')


def parse_list(input_list_str):
    return input_list_str.split(',')


argp = argparse.ArgumentParser()
argp.add_argument('function', help='Whether to pretrain, finetune or evaluate a model',
                  choices=['pretrain', 'finetune', 'embed', 'evaluate'])
argp.add_argument('name', help='Path to trained model weights (or "random" for a seeded random-init ViT).')
argp.add_argument('-l', '--load', help='Comma-separated list of processed dataset path.', default=None, type=parse_list)
argp.add_argument('-b', '--base', help='Path to base model.', default=None)
argp.add_argument('-s', '--sample', help='How many examples to sample for training.', default=None)
argp.add_argument('-a', '--auxiliary', action='store_true', default=False)
argp.add_argument('-t', '--test', help='Set flag to evaluate on test set.', action='store_true', default=False)
argp.add_argument('-c', '--classification', action='store_true', default=True)
argp.add_argument('-m', '--multitask', action='store_true', default=False)
argp.add_argument('--heading', action='store_true', default=False)
argp.add_argument('-r', '--resume', action='store_true', default=False)
argp.add_argument('--yfcc', action='store_true', default=False)
argp.add_argument('--landmarks', action='store_true', default=False)
# additions (not in the reference)
argp.add_argument('--synthetic', type=int, default=0, help='use N seeded synthetic panoramas instead of a dataset')
argp.add_argument('--layers', type=int, default=24, help='encoder layers for a "random" model')
argp.add_argument('--out-dir', default='data/landmark_embeddings')
argp.add_argument('--geocells', type=int, default=10000, help='synthetic geocell count')
argp.add_argument('--exact-top1', dest='exact_top1', action='store_true', default=None,
                  help="(default) re-encode the panoramas whose discrete outputs are inside the 16-bit path's error band in the "
                       "encoder's exact mode, so that geocell argmax and refined point are the reference's fp32 ones (PIGEON_EXACT_TOP1=1)")
argp.add_argument('--no-exact-top1', dest='exact_top1', action='store_false',
                  help="the 16-bit path alone (PIGEON_EXACT_TOP1=0): embeddings within 1e-3, discrete outputs not guaranteed")


class _SyntheticImages(torch.utils.data.Dataset):
    def __init__(self, n, panorama):
        self.n, self.panorama = n, panorama

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, str):                       # column access, as evaluate_model reads dataset['labels'] (train_eval_loop.py:123-124)
            import numpy as np
            return {'labels': np.zeros((self.n, 2)), 'labels_clf': np.zeros(self.n, dtype=np.int64)}[i]
        g = torch.Generator().manual_seed(1234 + i)
        if self.panorama:
            return {'pixel_values': torch.randn((12, 336, 336), generator=g), 'labels': torch.zeros(2, dtype=torch.float64),
                    'labels_clf': torch.tensor(0)}
        return {'image': torch.randn((3, 336, 336), generator=g), 'index': i}


def _vision_model(args):
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    path = args.base if args.function == 'evaluate' else args.name
    if path in (None, 'random') or not os.path.exists(str(path)):
        if path not in (None, 'random'):
            raise FileNotFoundError(path)
        logger.warning('Using a seeded random-init ViT-L/14-336 (%d layers): weights of the reference are not public.', args.layers)
        return HipCLIPVisionModel(seed=0, layers=args.layers)
    sd = torch.load(path, map_location='cpu')
    sd = {('.'.join(k.split('.')[1:]) if 'base_model' in k.split('.')[0] else k): v for k, v in sd.items()}
    return HipCLIPVisionModel(sd)


def main():
    args = argp.parse_args()
    if args.exact_top1 is not None:
        os.environ['PIGEON_EXACT_TOP1'] = '1' if args.exact_top1 else '0'     # read by SuperGuessr / HipCLIPVisionModel at construction
    mode = 'classification' if args.classification else 'regression'
    logger.warning(f'Task: {args.function.capitalize()} Pigeon("{args.name}") via geospatial {mode}.')
    if args.function in ('pretrain', 'finetune'):
        raise NotImplementedError(f'Mode {args.function} is training and not part of the MI355X inference hot path.')

    from pigeon_amd import distributed
    comm = distributed.init_from_env()
    try:
        res = _dispatch(args, comm)
    except BaseException:
        comm.close(rccl=False)  # a failing rank abandons its RCCL communicator (the others may sit in a collective)
        raise
    comm.barrier()
    comm.close()               # every rank: RCCL communicator + control-plane group, before the interpreter unwinds
    return res


def _dispatch(args, comm):
    from pigeon_amd import synthetic
    dev = f'cuda:{int(os.environ.get("LOCAL_RANK", "0"))}'
    torch.cuda.set_device(dev)

    if args.function == 'embed':
        if args.resume:
            raise NotImplementedError('Resuming from checkpoint not supported.')
        from pigeon_amd.clip_embedder import CLIPEmbedding
        from pigeon_amd.embed import embed_images
        embedder = CLIPEmbedding(args.name, device=dev, panorama=(not args.yfcc), clip_model=_vision_model(args))
        if args.synthetic:
            dataset = {'train': _SyntheticImages(args.synthetic, panorama=False)}
        else:
            from datasets import DatasetDict
            dataset = DatasetDict.load_from_disk(args.load[0])
        embed_images(embedder, dataset, comm, out_dir=args.out_dir, num_workers=0 if args.synthetic else 8)
        if comm.is_main_process:
            print(f'Embeddings written to {args.out_dir}/')
        return args.out_dir

    elif args.function == 'evaluate':
        from pigeon_amd.evaluate import evaluate
        import tempfile
        geocell_path = None
        bank = None
        if args.synthetic:
            tmp = tempfile.mkdtemp(prefix='pigeon_run_')
            geocell_path = os.path.join(tmp, 'geocells.csv')
            synthetic.write_geocell_csv(geocell_path, synthetic.make_geocells(args.geocells, seed=0))
            bank = synthetic.make_bank(args.geocells, 20, seed=2, exact_means=False)
            dataset = _SyntheticImages(args.synthetic, panorama=True)
        else:
            from datasets import DatasetDict
            dataset = DatasetDict.load_from_disk(args.load[0])
            if not args.yfcc:
                dataset = dataset['test'] if args.test else dataset['val']
        results = evaluate(args.name, dataset, yfcc=args.yfcc, base_model=_vision_model(args), refine=True,
                           landmarks=args.landmarks, geocell_path=geocell_path, bank=bank)
        if comm.is_main_process:
            print({k: (v if not hasattr(v, 'shape') or v.shape == () else tuple(v.shape)) for k, v in results.items() if k != 'exact_passes'})
            if 'geocell_certain' in results:
                # beyond the reference's output: how many samples' discrete outputs are certain to be the fp32 reference's (a z ~ 4
                # statistical statement, pigeon_amd/certainty.py), and how many are near-ties of the reference itself
                cert = results['geocell_certain']
                print(f"certain: {int(cert.sum())}/{cert.size} samples; still uncertain after the exact tier (returned as computed): "
                      f"{results.get('uncertain_after_exact', 0)}; exact passes: {[f['slots_run'] for f in results.get('exact_passes', [])]} panoramas")
        return results


if __name__ == '__main__':
    main()
