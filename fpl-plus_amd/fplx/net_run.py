"""`python -m fplx.net_run train|test config.cfg` - the reference's `pymic_run` entry for the DSBN agent
(PyMIC/pymic/net_run_dsbn/net_run.py:11-40): parse + synchronize the .cfg, log to <ckpt_save_dir>/log_<stage>.txt,
SegmentationAgent(config, stage).run(); after a training stage the test stage runs with the best checkpoint, then the
evaluation reports are written (util/evaluation_seg_train.py:577-582: evaluation_1 with metric_1, evaluation_2 with
metric_2 - assd in every shipped cfg)."""
import logging
import os
import sys

from .agent import SegmentationAgent
from .config import parse_config, synchronize_config
from . import evaluation


def eva_main(config):
    if 'evaluation' not in config:
        return None
    res = evaluation.evaluation_1(config)
    if config['evaluation'].get('metric_2', None) is not None:
        evaluation.evaluation_2(config)
    return res


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 3:
        print('Number of arguments should be 3. e.g.')
        print('   python -m fplx.net_run train config.cfg')
        return 1
    stage, cfg_file = str(argv[1]), str(argv[2])
    config = synchronize_config(parse_config(cfg_file))
    log_dir = config['training']['ckpt_save_dir']
    os.makedirs(log_dir, exist_ok=True)
    logging.basicConfig(filename=log_dir + "/log_{0:}.txt".format(stage), level=logging.INFO, format='%(message)s',
                        force=True)
    task = config['dataset'].get('task_type', 'seg')
    if task != 'seg':
        raise ValueError("fplx.net_run: only task_type = seg is built (got {0:})".format(task))
    from . import ddp
    agent = SegmentationAgent(config, stage)
    agent.run()
    if stage != 'test':
        ddp.barrier()                      # rank 0 has written the best checkpoint
        agent2 = SegmentationAgent(config, 'test')
        agent2.run()
    ddp.barrier()                          # every rank has written its share of the predictions
    return eva_main(config) if ddp.rank() == 0 else None


if __name__ == "__main__":
    r = main()
    sys.exit(r if isinstance(r, int) else 0)
