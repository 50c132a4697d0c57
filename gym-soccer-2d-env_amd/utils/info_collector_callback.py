"""Drop-in for ``utils.info_collector_callback.InfoCollectorCallback`` (reference
info_collector_callback.py:5-78): tallies info['result'] of finished episodes in blocks of 100
and reports Goal / Out / Timeout percentages.  Works as an SB3 callback when SB3 is installed,
and as a plain object (call ``on_infos(infos)`` or ``on_result_codes(tensor)``) otherwise."""
try:  # pragma: no cover
    from stable_baselines3.common.callbacks import BaseCallback as _Base  # type: ignore
except Exception:  # noqa: BLE001
    class _Base:  # minimal stand-in with the attributes _on_step reads
        def __init__(self):
            self.locals = {}

RESULT_TYPES = ('Goal', 'Out', 'Timeout')


class InfoCollectorCallback(_Base):
    def __init__(self):
        super().__init__()
        self.infos = []
        self.results = {}

    def _on_step(self):
        infos = self.locals.get('infos')
        if infos is not None:
            self.on_infos(infos)
        return True

    def on_infos(self, infos):
        finished = infos.finished() if hasattr(infos, 'finished') else infos
        for info in finished:
            if info['result'] and len(info['result']) > 0:
                self.infos.append(info)

    def on_result_codes(self, codes):
        """Device-side variant: `codes` is the uint8 result tensor of a step or rollout."""
        import torch
        c = codes.flatten()
        c = c[c != 0].cpu().tolist() if torch.is_tensor(c) else [x for x in c if x]
        names = (None,) + RESULT_TYPES
        self.infos.extend({'result': names[x]} for x in c)

    def reset(self):
        self.infos = []
        self.results = {}

    def update_results_dict(self, logger=None):
        results = [info['result'] for info in self.infos]
        self.results = {t: [] for t in RESULT_TYPES}
        for i in range(0, len(results), 100):
            block = results[i:i + 100]
            for t in RESULT_TYPES:
                self.results[t].append(block.count(t) / len(block) * 100)
        if logger is not None:
            logger.info(f"Results dictionary: {self.results}")
        return self.results

    def plot_print_results(self, logger=None, file_name=None):
        self.update_results_dict(logger)
        try:
            import matplotlib
            matplotlib.use('Agg')
            import matplotlib.pyplot as plt
            fig, ax = plt.subplots()
            for t in self.results:
                ax.plot(self.results[t], label=t)
            ax.legend(); ax.set_xlabel('Episodes (x100)'); ax.set_ylabel('Percentage'); ax.set_title('Results')
            if file_name:
                plt.savefig(file_name + '.png')
            plt.close(fig)
        except Exception:  # noqa: BLE001  plotting is optional
            pass
        return self.results['Goal'], self.results['Out'], self.results['Timeout']
